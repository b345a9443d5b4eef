"""starvector_amd: MI355X (gfx950)-native StarVector im2svg inference engine.

The directory is named ``star-vector_amd`` (repo convention); import it as ``starvector_amd`` (the
sibling shim package maps the importable name onto this directory).
"""
from ._lib import StarVectorHipError, LIB_PATH, HEADER_PATH  # noqa: F401
from .engine import EngineConfig, HipEngine  # noqa: F401
from .model import (  # noqa: F401
    StarVectorConfig, StarVectorForCausalLM, StarVectorStarCoder, StarVectorStarCoder2, StarCoderModel, ImageEncoder, Adapter,
    HipCausalLM, StoppingCriteriaSub, ImageTrainProcessor, SimpleStarVectorProcessor, ByteTokenizer,
)
from .batching import ContinuousBatcher  # noqa: F401
from . import parallel  # noqa: F401

__all__ = ["EngineConfig", "HipEngine", "StarVectorConfig", "StarVectorForCausalLM", "StarVectorStarCoder", "StarVectorStarCoder2",
           "StarCoderModel", "ImageEncoder", "Adapter", "HipCausalLM", "StoppingCriteriaSub",
           "ImageTrainProcessor", "SimpleStarVectorProcessor", "ByteTokenizer", "StarVectorHipError", "ContinuousBatcher", "parallel"]
