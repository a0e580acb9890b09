"""Drop-in counterparts of the reference's src/extractor modules (same function names and
argument meaning), running on the HIP engine."""
from . import visualise_resnet, visualise_resnet_layer, visualise_vit_layer  # noqa: F401
