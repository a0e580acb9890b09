"""BASELINE config 5 on the GPU: 3840x2160 clips, 32 pairs, full ReLaX (residual + optical-flow fragments, whole-frame and
fragment features of both backbones -> 35203-d).  Reference path: src/demo_test.py:76-175, src/main_fragment_layerstack.py:313-325.
Round-1 review: only stage A and the resize had run at 2160p."""
import numpy as np
import pytest
import torch

from oracle import flow_ref, fragment_ref, resize_ref, resnet50_ref, vit_ref
from tests.gpu_common import assert_close, engine, rn50_weights, synth, vit_weights
from tests.test_gpu_flow import _smooth_pair

pytestmark = pytest.mark.gpu
H, W = 2160, 3840


def test_farneback_at_2160p_against_the_oracle():
    """One 2160x3840 pair: the flow itself against oracle/flow_ref.farneback, same bars as at 540p / 1080p."""
    a, b = _smooth_pair(H, W, 7)
    flow, img = engine().optical_flow(torch.from_numpy(np.stack([a, b])[None]).cuda(), want_flow=True, want_image=True)
    want = flow_ref.farneback(flow_ref.bgr2gray(a), flow_ref.bgr2gray(b))
    err = np.abs(flow[0].cpu().numpy() - want)
    assert err.max() < 2e-3 and err.mean() < 1e-5, (err.max(), err.mean())
    want_img = flow_ref.flow_to_rgb(want)
    d = np.abs(img[0].cpu().numpy().astype(np.int32) - want_img.astype(np.int32))
    assert (d == 0).mean() > 0.995 and (d <= 1).mean() > 0.9995, ((d == 0).mean(), (d <= 1).mean())


def test_flow_chunk_loop_is_taken_and_changes_nothing():
    """relax_optical_flow walks a clip in chunks of pairs when the workspace is capped (csrc/flow.hip, the chunk loop): three
    pairs forced into chunks of 2 + 1 and of 1 + 1 + 1 give the bits of the unchunked run, flow and image."""
    eng = engine()
    pairs = [_smooth_pair(360, 640, s) for s in (11, 12, 13)]
    frames = torch.from_numpy(np.stack([np.stack(p) for p in pairs])).cuda()
    assert eng.get_option("flow_max_pairs") == 0
    flow0, img0 = eng.optical_flow(frames, want_flow=True, want_image=True)
    try:
        for cap in (2, 1):
            eng.set_option("flow_max_pairs", cap)
            flow1, img1 = eng.optical_flow(frames, want_flow=True, want_image=True)
            assert torch.equal(flow1, flow0) and torch.equal(img1, img0), f"chunks of {cap} pairs changed the result"
    finally:
        eng.set_option("flow_max_pairs", 0)


def test_config5_two_clips_of_32_pairs_deterministic_and_batch_invariant():
    """2 clips x 32 pairs of 2160p through full_clip_vectors (flow on the GPU): two runs agree bit for bit; with the tail split
    off a clip gets the bits it gets alone; with a workspace cap that forces the flow chunk loop the bits stay."""
    rn50_weights(), vit_weights("vit_base")
    eng = engine()
    clips = [torch.from_numpy(synth.synthetic_clip(32, H, W, clip_id=500 + i, distinct=2)).cuda() for i in range(2)]
    eng.set_option("gemm_split_k", 0)
    try:
        both = eng.full_clip_vectors(clips, flow=True)
        again = eng.full_clip_vectors(clips, flow=True)
        assert both.shape == (2, 35203) and bool(torch.isfinite(both).all())
        assert torch.equal(both, again), "config 5 is not deterministic"
        alone = eng.full_clip_vectors([clips[1]], flow=True)
        assert torch.equal(alone[0], both[1]), "a config-5 clip depends on its batch"
        eng.set_option("flow_max_pairs", 12)          # 32 pairs -> chunks of 12 + 12 + 8
        chunked = eng.full_clip_vectors([clips[1]], flow=True)
        assert torch.equal(chunked[0], both[1]), "the flow chunk loop changed a config-5 vector"
    finally:
        eng.set_option("gemm_split_k", 1)
        eng.set_option("flow_max_pairs", 0)
    split = eng.full_clip_vectors(clips, flow=True)
    assert_close(split, both.cpu().numpy(), "tail split-K on vs off at config 5", rtol=1e-4, atol_frac=1e-5)


def test_config5_every_block_of_the_35203_vector_against_the_oracle(each_precision, each_split_k):
    """One 2160p pair, block by block against the oracle: whole-frame RN50 layer stack | whole-frame ViT | fragment layer stack
    | residual(+flow) pool | ViT of both fragments.  The flow images come from the GPU (checked against the oracle above) and
    go to both sides, so the comparison does not hinge on a near-tie between two flow patches."""
    rn, vit = rn50_weights(), vit_weights("vit_base")
    eng = engine()
    clip = synth.synthetic_clip(1, H, W, clip_id=521)
    frames = torch.from_numpy(clip).cuda()
    _, flow_img = eng.optical_flow(frames)
    vec = eng.full_clip_vector(frames, flow_images=flow_img).cpu().numpy()
    fimg = flow_img.cpu().numpy()
    tr, tv = resnet50_ref.to_torch_state_dict(rn), vit_ref.to_torch_state_dict(vit)
    fp = fragment_ref.fragment_pair(clip[0, 0], clip[0, 1])
    flow_frag, _ = fragment_ref.extract_important_patches(fimg[0], fragment_ref.get_patch_diff(fimg[0]))
    ori = fp["ori_frag"][None]
    res = fragment_ref.merge_fragments(fp["diff_frag"], flow_frag)[None]
    bil = resize_ref.resize(clip[0, 0], 224, 224, resize_ref.BILINEAR)[None]
    lan = resize_ref.resize(clip[0, 0], 224, 224, resize_ref.LANCZOS)[None]
    want = np.concatenate([
        resnet50_ref.layer_stack_features(tr, bil)[0], vit_ref.pool_features(tv, lan, 12)[0],
        resnet50_ref.layer_stack_features(tr, ori)[0], resnet50_ref.pool_features(tr, res)[0],
        vit_ref.pool_features(tv, ori, 12)[0], vit_ref.pool_features(tv, res, 12)[0]])
    assert want.shape == (35203,)
    edges = [0, 13120, 15424, 28544, 30595, 32899, 35203]
    names = ["whole-frame RN50 LS", "whole-frame ViT", "fragment RN50 LS", "residual RN50 pool", "ViT original frag", "ViT residual frag"]
    for a, b, nm in zip(edges[:-1], edges[1:], names):
        assert_close(vec[a:b], want[a:b], f"config 5 block: {nm} ({each_precision})")


def test_whole_frames_of_an_unpaired_last_sample_enter_the_whole_frame_mean():
    """src/demo_test.py:76-87 averages the whole-frame features over EVERY sampled frame; when n_frames % k == 1 the last
    sampled frame has no `next` partner, so it is in no pair but still in that mean (whole_frames argument)."""
    rn50_weights(), vit_weights("vit_base")
    eng = engine()
    clip = synth.synthetic_clip(3, 272, 400, clip_id=31)
    frames = torch.from_numpy(clip[:2]).cuda()                       # two pairs ...
    sampled = torch.from_numpy(np.ascontiguousarray(clip[:, 0])).cuda()   # ... but three sampled frames
    eng.set_option("gemm_split_k", 0)                                # bits independent of how many images travel together
    try:
        a = eng.full_clip_vector(frames, whole_frames=sampled)
        b = eng.full_clip_vector(frames)
        ls, vp = eng.whole_frame_features(sampled)
    finally:
        eng.set_option("gemm_split_k", 1)
    assert torch.equal(a[15424:], b[15424:])                         # fragment blocks: pairs only
    assert_close(a[:13120], ls.mean(dim=0).cpu().numpy(), "whole-frame RN50 mean over 3 frames", rtol=1e-4, atol_frac=1e-5)
    assert_close(a[13120:15424], vp.mean(dim=0).cpu().numpy(), "whole-frame ViT mean over 3 frames", rtol=1e-4, atol_frac=1e-5)
    assert not torch.equal(a[:15424], b[:15424])
