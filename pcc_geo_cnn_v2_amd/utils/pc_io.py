"""PLY I/O + shape helpers -- the interface of /root/reference/src/utils/pc_io.py:17-79 without
pyntcloud: a small self-contained PLY reader/writer (ascii and binary_little_endian, `vertex` element).
"""
import logging
import multiprocessing
import os

import numpy as np
import pandas as pd

logger = logging.getLogger(__name__)

_PLY_TYPES = {'char': 'i1', 'int8': 'i1', 'uchar': 'u1', 'uint8': 'u1', 'short': 'i2', 'int16': 'i2',
              'ushort': 'u2', 'uint16': 'u2', 'int': 'i4', 'int32': 'i4', 'uint': 'u4', 'uint32': 'u4',
              'float': 'f4', 'float32': 'f4', 'double': 'f8', 'float64': 'f8'}
_NP_TO_PLY = {'float32': 'float', 'float64': 'double', 'uint8': 'uchar', 'int8': 'char', 'int16': 'short',
              'uint16': 'ushort', 'int32': 'int', 'uint32': 'uint'}


def read_ply(path):
    """Returns a DataFrame with the vertex properties of a PLY file."""
    with open(path, 'rb') as f:
        assert f.readline().strip() == b'ply', f'{path} is not a PLY file'
        fmt, props, n_vertex, in_vertex = None, [], 0, False
        other_elements_before = False
        while True:
            line = f.readline()
            assert line, f'{path}: unexpected end of header'
            tok = line.decode('ascii', 'replace').split()
            if not tok:
                continue
            if tok[0] == 'format':
                fmt = tok[1]
            elif tok[0] == 'element':
                in_vertex = tok[1] == 'vertex'
                if in_vertex:
                    n_vertex = int(tok[2])
                elif n_vertex == 0:
                    other_elements_before = True
            elif tok[0] == 'property' and in_vertex:
                assert tok[1] != 'list', 'list properties on vertices are not supported'
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == 'end_header':
                break
        assert not other_elements_before, 'vertex must be the first element'
        if fmt == 'ascii':
            data = np.loadtxt(f, max_rows=n_vertex, ndmin=2) if n_vertex else np.zeros((0, len(props)))
            return pd.DataFrame({name: data[:, i].astype(dt) for i, (name, dt) in enumerate(props)})
        endian = '<' if fmt == 'binary_little_endian' else '>'
        dtype = np.dtype([(name, endian + dt) for name, dt in props])
        data = np.frombuffer(f.read(dtype.itemsize * n_vertex), dtype=dtype, count=n_vertex)
        return pd.DataFrame({name: data[name].astype(data[name].dtype.newbyteorder('=')) for name, _ in props})


def write_ply(path, df, as_text=False):
    cols = list(df.columns)
    n = len(df)
    header = ['ply', 'format ascii 1.0' if as_text else 'format binary_little_endian 1.0', f'element vertex {n}']
    dts = []
    for c in cols:
        dt = np.dtype(df[c].dtype)
        name = _NP_TO_PLY.get(dt.name)
        if name is None:
            dt, name = np.dtype('float32'), 'float'
        dts.append(dt)
        header.append(f'property {name} {c}')
    header.append('end_header')
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, 'wb') as f:
        f.write(('\n'.join(header) + '\n').encode('ascii'))
        if as_text:
            np.savetxt(f, np.column_stack([df[c].values for c in cols]) if n else np.zeros((0, len(cols))), fmt='%g')
        else:
            rec = np.empty(n, dtype=np.dtype([(c, '<' + dt.str[1:]) for c, dt in zip(cols, dts)]))
            for c in cols:
                rec[c] = df[c].values
            f.write(rec.tobytes())


def df_to_pc(df):
    return df[['x', 'y', 'z']].values


def pa_to_df(points):
    cols = ['x', 'y', 'z', 'red', 'green', 'blue']
    types = (['float32'] * 3) + (['uint8'] * 3)
    assert 3 <= points.shape[1] <= 6
    return pd.DataFrame(data={cols[i]: points[:, i].astype(types[i]) for i in range(points.shape[1])})


def load_pc(path):
    logger.debug(f'Loading PC {path}')
    return df_to_pc(read_ply(path))


def load_normals(path):
    return read_ply(path)[['nx', 'ny', 'nz']].values


def write_pc(path, pc):
    write_df(path, pa_to_df(pc))


def write_df(path, df):
    write_ply(path, df)


def get_shape_data(resolution, data_format):
    assert data_format in ['channels_last', 'channels_first']
    p_max = np.array([resolution, resolution, resolution])
    p_min = np.array([0, 0, 0])
    if data_format == 'channels_last':
        dense_tensor_shape = np.concatenate([p_max, [1]]).astype('int64')
    else:
        dense_tensor_shape = np.concatenate([[1], p_max]).astype('int64')
    return p_min, p_max, dense_tensor_shape


def load_points(files, batch_size=32):
    files = list(files)
    if len(files) <= 1:
        return [load_pc(f) for f in files]
    with multiprocessing.Pool() as p:
        logger.info('Loading PCs into memory (parallel reading)')
        return list(p.imap(load_pc, files, batch_size))
