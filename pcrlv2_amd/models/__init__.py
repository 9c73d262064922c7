"""`from models import PCRLv23d` of the reference (models/__init__.py) -- the 3D model does not depend on
segmentation_models_pytorch here."""
from .pcrlv2_model_3d import PCRLv23d  # noqa: F401
