from .autoencoder_kl import AutoencoderKL  # noqa: F401
from .modeling_utils import ModelMixin  # noqa: F401
