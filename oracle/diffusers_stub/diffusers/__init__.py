"""Test-only stub of `diffusers==0.24.0` — see ../README.md.  NOT the product."""
__version__ = "0.24.0+stub"

from .models import AutoencoderKL, ModelMixin  # noqa: F401
from .pipeline_utils import DiffusionPipeline  # noqa: F401
from .schedulers import DDIMScheduler  # noqa: F401
