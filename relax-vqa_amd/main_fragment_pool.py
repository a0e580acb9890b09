"""Counterpart of the reference driver src/main_fragment_pool.py (fragments + `pool` features of one backbone, the
driver that produces the ViT half of the fragment features): same names, argument order and return arity, on the HIP
engine.  It differs from main_fragment_layerstack only in get_deep_feature's layer names ('pool' | 'last_layer',
reference :83-111) and the two-argument process_video_feature (:114-143)."""
from .main_fragment_layerstack import (concatenate_features, extract_important_patches, flow_to_rgb,  # noqa: F401
                                       fragment_pair, get_original_frame_patches, get_patch_diff, merge_fragments,
                                       process_patches)
from .main_residual_fragment import get_deep_feature, process_video_feature  # noqa: F401
