"""`from models import PCRLv2, PCRLv23d` of the reference (models/__init__.py) -- neither model depends on
segmentation_models_pytorch here."""
from .pcrlv2_model import PCRLv2  # noqa: F401
from .pcrlv2_model_3d import PCRLv23d  # noqa: F401
