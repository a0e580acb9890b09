"""Oracle (TEST INFRASTRUCTURE, not product): numpy restatement of the integer
fragment path of ReLaX-VQA.

Follows, in /root/reference/src:
  main_fragment_layerstack.py:302       cv2.absdiff               -> absdiff
  main_fragment_layerstack.py:177-189   get_patch_diff            -> get_patch_diff / get_patch_diff_loop
  main_fragment_layerstack.py:191-210   extract_important_patches -> select_positions + extract_important_patches
  main_fragment_layerstack.py:212-230   get_original_frame_patches
  main_fragment_layerstack.py:242-245   merge_fragments (cv2.addWeighted .5/.5)
  main_residual_fragment.py:173-214     same functions, PNG-writing variant

Everything here is exact integer arithmetic; the HIP path must match bit for
bit.  Tie rule: the reference uses ``np.argsort(-diff)``
(unstable); only a tie straddling rank top_n/top_n+1 makes the selected *set*
implementation-defined.  Oracle and HIP both use "higher score first, then
lower flat index", i.e. a stable descending sort.
"""
import numpy as np


def absdiff(img_next, img_original):
    """|next - orig| per byte, uint8 (cv2.absdiff)."""
    a = img_next.astype(np.int16)
    b = img_original.astype(np.int16)
    return np.abs(a - b).astype(np.uint8)


def get_patch_diff(residual_frame, patch_size=16):
    """Per patch sum of all bytes (np.abs is the identity on uint8; np.sum
    promotes to uint64; the reference stores float64).  Rows/cols beyond the
    last whole patch are ignored."""
    h, w = residual_frame.shape[:2]
    ph, pw = h // patch_size, w // patch_size
    r = residual_frame[:ph * patch_size, :pw * patch_size]
    r = r.reshape(ph, patch_size, pw, patch_size, residual_frame.shape[2]).astype(np.uint64)
    return r.sum(axis=(1, 3, 4)).astype(np.float64)


def get_patch_diff_loop(residual_frame, patch_size=16):
    """Same result, computed the way the reference does (one python iteration
    per patch); used only by the reference-faithful CPU baseline schedule."""
    h, w = residual_frame.shape[:2]
    ph, pw = h // patch_size, w // patch_size
    out = np.zeros((ph, pw))
    for py in range(ph):
        y0 = py * patch_size
        for px in range(pw):
            x0 = px * patch_size
            out[py, px] = np.sum(np.abs(residual_frame[y0:y0 + patch_size, x0:x0 + patch_size]))
    return out


def select_positions(diff, top_n=196):
    """Top-n patches by score (desc), ties by lower flat index, returned in
    raster order.  -> int32 [n, 2] of (y, x), n = min(top_n, diff.size)."""
    flat = diff.ravel()
    order = np.argsort(-flat, kind="stable")[:top_n]
    order = np.sort(order)
    ys, xs = np.unravel_index(order, diff.shape)
    return np.stack([ys, xs], axis=1).astype(np.int32)


def gather_patches(frame, positions, patch_size=16, target_size=224):
    """Copy patch k=(y,x) to tile (k // 14, k % 14) of a zeroed canvas."""
    canvas = np.zeros((target_size, target_size, frame.shape[2]), dtype=frame.dtype)
    per_row = target_size // patch_size
    for k, (y, x) in enumerate(positions):
        ty, tx = divmod(k, per_row)
        canvas[ty * patch_size:(ty + 1) * patch_size, tx * patch_size:(tx + 1) * patch_size] = \
            frame[y * patch_size:(y + 1) * patch_size, x * patch_size:(x + 1) * patch_size]
    return canvas


def extract_important_patches(residual_frame, diff, patch_size=16, target_size=224, top_n=196):
    positions = select_positions(diff, top_n)
    return gather_patches(residual_frame, positions, patch_size, target_size), positions


def get_original_frame_patches(original_frame, positions, patch_size=16, target_size=224):
    return gather_patches(original_frame, positions, patch_size, target_size)


def merge_fragments(diff_fragment, flow_fragment):
    """cv2.addWeighted(a, .5, b, .5, 0) on uint8 == round-half-to-even(0.5a+0.5b)
    (verified bit-exact on the reference's four example PNG sets)."""
    s = diff_fragment.astype(np.float64) * 0.5 + flow_fragment.astype(np.float64) * 0.5
    return np.rint(s).astype(np.uint8)


def fragment_pair(img_original, img_next, patch_size=16, target_size=224, top_n=196, loop_score=False):
    """One (frame, next) pair: residual -> score -> positions -> (diff_frag, ori_frag)."""
    residual = absdiff(img_next, img_original)
    diff = (get_patch_diff_loop if loop_score else get_patch_diff)(residual, patch_size)
    diff_frag, positions = extract_important_patches(residual, diff, patch_size, target_size, top_n)
    ori_frag = get_original_frame_patches(img_original, positions, patch_size, target_size)
    return dict(score=diff, positions=positions, diff_frag=diff_frag, ori_frag=ori_frag)
