"""Named model configurations -- /root/reference/src/model_configs.py:7-49 (c1, c2, c3, c3p), plus
aliases for the experiment ids / paper labels of src/ev_experiment.yml:10-46:
the paper's c3..c6 are all the `c3p` network (they differ in trained weights and threshold policy).
"""
from enum import Enum

from .model_transforms import TransformType
from .model_types import ModelType


class ModelConfig:
    def __init__(self, model_type: ModelType, model_params):
        self.model_type = model_type
        self.model_params = model_params

    def build(self, **overrides):
        return self.model_type.value(**{**self.model_params, **overrides})


class ModelConfigType(Enum):
    c1 = ModelConfig(ModelType.v1, {
        'num_filters': 32,
        'analysis_transform_type': TransformType.AnalysisTransformV1,
        'synthesis_transform_type': TransformType.SynthesisTransformV1
    })
    c2 = ModelConfig(ModelType.v2, {
        'num_filters': 32,
        'analysis_transform_type': TransformType.AnalysisTransformV1,
        'synthesis_transform_type': TransformType.SynthesisTransformV1,
        'hyper_analysis_transform_type': TransformType.HyperAnalysisTransform,
        'hyper_synthesis_transform_type': TransformType.HyperSynthesisTransform
    })
    c3 = ModelConfig(ModelType.v2, {
        'num_filters': 32,
        'analysis_transform_type': TransformType.AnalysisTransformV2,
        'synthesis_transform_type': TransformType.SynthesisTransformV2,
        'hyper_analysis_transform_type': TransformType.HyperAnalysisTransform,
        'hyper_synthesis_transform_type': TransformType.HyperSynthesisTransform
    })
    c3p = ModelConfig(ModelType.v2, {
        'num_filters': 64,
        'analysis_transform_type': TransformType.AnalysisTransformProgressiveV2,
        'synthesis_transform_type': TransformType.SynthesisTransformProgressiveV2,
        'hyper_analysis_transform_type': TransformType.HyperAnalysisTransform,
        'hyper_synthesis_transform_type': TransformType.HyperSynthesisTransform
    })

    @staticmethod
    def keys():
        return ModelConfigType.__members__.keys()

    def build(self, **overrides):
        return self.value.build(**overrides)


# experiment id / paper label -> (architecture key, fixed_threshold) -- src/ev_experiment.yml:10-46,53
EXPERIMENT_ALIASES = {
    'c1': ('c1', True), 'c2': ('c2', True), 'c3p': ('c3p', True), 'c3p-a0.75': ('c3p', True),
    'c3p-a0.5': ('c3p', True), 'c3p-a0.25': ('c3p', True), 'c4': ('c3p', False), 'c4-ws': ('c3p', False),
}
PAPER_LABELS = {'c1': 'c1', 'c2': 'c2', 'c3': 'c3p', 'c4': 'c3p-a0.75', 'c5': 'c4', 'c6': 'c4-ws'}
