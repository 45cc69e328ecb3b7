"""hirest_amd — MI355X-native implementation of HiREST's frame/text encoding + cross-modal
scoring hot path.  Host side mirrors the reference's Python interface; all
compute is in libhirest_hip.so (include/hirest_hip.h)."""
from .eva_clip import (EVA_CLIP, build_eva_model_and_transforms, create_model, image_transform,  # noqa: F401
                       get_model_config, list_models)
from .tokenizer import tokenize  # noqa: F401
from .moment_model import MomentModel  # noqa: F401
from .sentence_encoder import SentenceTransformer  # noqa: F401

__version__ = "0.1.0"
