"""Opt-in phase stamps of the two command lines (tools/cli_wallclock.py).  With PCC_CLI_TIMING_JSON=<path> in the environment,
compress_octree / decompress_octree append (phase name, time.time()) pairs and write them to <path> on exit; without it every
call is a no-op.  The stamps bracket what a user of the drop-in waits for around the measured block loop: interpreter + imports,
context creation, PLY parse, octree partition, checkpoint restore + weight repack / upload, the codec calls, container / PLY writes
(/root/reference/src/compress_octree.py:36-127, decompress_octree.py:30-145)."""
import json
import os
import time

_PATH = os.environ.get('PCC_CLI_TIMING_JSON')
_marks = []


def enabled():
    return _PATH is not None


def mark(name, sync=None):
    """Stamp the END of phase `name`; `sync` = a torch device whose queue is drained first (GPU work belongs to the phase that issued it)."""
    if _PATH is None:
        return
    if sync is not None:
        import torch
        torch.cuda.synchronize(sync)
    _marks.append((name, time.time()))


def dump():
    if _PATH is not None:
        with open(_PATH, 'w') as fh:
            json.dump({'marks': _marks, 'pid': os.getpid()}, fh)
