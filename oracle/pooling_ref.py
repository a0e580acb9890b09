"""Oracle (TEST INFRASTRUCTURE, not product): numpy restatement of the feature
pooling of ReLaX-VQA.

Follows, in /root/reference/src:
  main_fragment_layerstack.py:124-160  process_video_feature(video_feature, network_name, layer_name)
                                       'layer_stack' branch :134-140, 'pool' branch :141-149
  main_residual_fragment.py:118-156    process_video_feature(video_feature, network_name), vit branch :128-136
  main_fragment_pool.py:114-143        vit-only variant
  main_fragment_layerstack.py:247-248  concatenate_features
  demo_test.py:171-175                 per-clip mean over frames
"""
import numpy as np

RESNET50_TAPS = [
    "resnet50.conv1",
    "resnet50.layer1[0]", "resnet50.layer1[1]", "resnet50.layer1[2]",
    "resnet50.layer2[0]", "resnet50.layer2[1]", "resnet50.layer2[2]", "resnet50.layer2[3]",
    "resnet50.layer3[0]", "resnet50.layer3[1]", "resnet50.layer3[2]", "resnet50.layer3[3]",
    "resnet50.layer4[0]", "resnet50.layer4[1]", "resnet50.layer4[2]",
]
RESNET50_TAP_CHANNELS = [64, 256, 256, 256, 512, 512, 512, 512, 1024, 1024, 1024, 1024, 2048, 2048, 2048]
LAYER_STACK_DIM = sum(RESNET50_TAP_CHANNELS)  # 13120
RESNET50_POOL_DIM = 2048 + 3                  # 2051
VIT_POOL_DIM = 3 * 768                        # 2304


def layer_stack_vector(frame_taps):
    """dict/list of [C,H,W] fp32 arrays (dict order) -> [sum C] spatial means."""
    taps = frame_taps.values() if isinstance(frame_taps, dict) else frame_taps
    return np.hstack([np.mean(t, axis=(1, 2)) for t in taps])


def resnet_pool_vector(avgpool_out):
    """[2048,1,1] -> [2051]: the vector plus its scalar mean, max, population std."""
    v = np.squeeze(avgpool_out)
    return np.hstack([v, np.mean(v, axis=0), np.max(v, axis=0), np.std(v, axis=0)])


def vit_pool_vector(tokens):
    """[196,768] -> [2304]: per-channel mean, max, population std over tokens."""
    return np.hstack([np.mean(tokens, axis=0), np.max(tokens, axis=0), np.std(tokens, axis=0)])


def process_video_feature(video_feature, network_name, layer_name="pool"):
    rows = []
    for frame in video_feature:
        if network_name == "vit":
            rows.append(vit_pool_vector(frame))
        elif layer_name == "layer_stack":
            rows.append(layer_stack_vector(frame))
        else:
            rows.append(resnet_pool_vector(frame))
    return np.array(rows)


def concatenate_features(original_feature, residual_feature):
    return np.concatenate((original_feature, residual_feature), axis=-1)


def clip_mean(per_frame_features):
    return np.mean(per_frame_features, axis=0)
