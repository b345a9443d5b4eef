"""starvector_amd: MI355X (gfx950)-native StarVector im2svg inference engine.

The directory is named ``star-vector_amd`` (repo convention); import it as ``starvector_amd`` (the
sibling shim package maps the importable name onto this directory).
"""
import os as _os

# Multi-process GPU work (generate_im2svg_dp: one process per GPU, RCCL all-gather) needs dmabuf IPC on hosts whose driver has no
# legacy IPC -- without this RCCL's hipIpcGetMemHandle fails with "invalid argument".  The HSA runtime reads the variable ONCE, when
# torch first initialises HIP, so it has to be in the environment before that.  The package sets it ONLY in a process that a
# distributed launcher started (torchrun / torch.distributed.run export WORLD_SIZE): a plain single-GPU import changes nothing in the
# caller's environment; an explicit value from the launcher wins (setdefault); SV_LOG_ENV=1 says when it was applied.  If HIP
# is already up in this process the variable must come from the launcher's environment instead.
if int(_os.environ.get("WORLD_SIZE", "1") or "1") > 1 and "HSA_ENABLE_IPC_MODE_LEGACY" not in _os.environ:
    _os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if _os.environ.get("SV_LOG_ENV"):
        import sys as _sys
        print("[starvector_amd] WORLD_SIZE > 1: set HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC for RCCL)", file=_sys.stderr)

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
