import os

import numpy as np

from oracle import pooling_ref


def test_pooling_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "pooling.npz"))
    n_frames = z["ls_expected"].shape[0]
    frames = []
    for i in range(n_frames):
        keys = sorted(k for k in z.files if k.startswith(f"ls_in/{i}/"))
        frames.append({n: z[k] for n, k in zip(pooling_ref.RESNET50_TAPS, keys)})
    assert np.array_equal(pooling_ref.process_video_feature(frames, "resnet50", "layer_stack"), z["ls_expected"])
    assert np.array_equal(pooling_ref.process_video_feature(list(z["pool_in"]), "resnet50", "pool"), z["pool_expected"])
    assert np.array_equal(pooling_ref.process_video_feature(list(z["vit_in"]), "vit"), z["vit_expected"])


def test_dims():
    assert pooling_ref.LAYER_STACK_DIM == 13120
    assert pooling_ref.LAYER_STACK_DIM + pooling_ref.RESNET50_POOL_DIM == 15171
    # 35203 = whole-frame RN50-LS + whole-frame ViT + fragment RN50 (LS+pool) + fragment ViT (2x)
    assert 13120 + 2304 + 15171 + 2 * 2304 == 35203
