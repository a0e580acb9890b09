"""Host side of the MI355X feature-extraction engine.

PyTorch-ROCm is plumbing only (device memory allocation, streams, torch.distributed, tiny host-to-device copies of
index vectors); all compute of the clip paths (clip_vectors / full_clip_vectors: fragments, backbones, per-clip means) goes
through the C-ABI of librelax_hip.so (include/relax_hip.h) - the fragments are written straight into the batch buffer the
backbones read and the per-clip means straight into the result matrix, so no aten kernel runs in between.
Array-in / array-out counterparts of the reference's path-based functions:

  fragment_pairs      <- cv2.absdiff + process_patches('frame_diff') + get_original_frame_patches
                         (src/main_fragment_layerstack.py:302-310)
  fragment_image      <- process_patches('optical_flow', flow_rgb) (src/main_fragment_layerstack.py:319)
  merge_fragments     <- merge_fragments (src/main_fragment_layerstack.py:242-245)
  resnet50_features   <- get_deep_feature('resnet50', .., 'layer_stack'|'pool') + process_video_feature
                         (src/main_fragment_layerstack.py:83-99,124-160)
  vit_features        <- get_deep_feature('vit', ..) + process_video_feature (src/main_fragment_pool.py:114-143)
  extract_clip        <- the per-video loop body of src/main_fragment_layerstack.py:293-344 plus the ViT
                         branch of src/demo_test.py:137-161 (config 3 of BASELINE.json)
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

LAYER_STACK_DIM = 13120
RN50_POOL_DIM = 2051
TOP_N = 196
TARGET = 224
RN50_TAP_SHAPES = [(64, 112)] + [(256, 56)] * 3 + [(512, 28)] * 4 + [(1024, 14)] * 4 + [(2048, 7)] * 3
VIT_CONFIGS = {"vit_tiny": (192, 12, 3), "vit_small": (384, 12, 6), "vit_base": (768, 12, 12)}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class RelaxEngine:
    """One engine per (process, device).  Not thread-safe (mirrors the C handle)."""

    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("RelaxEngine needs a ROCm GPU (MI355X); there is no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        torch.cuda.set_device(self.device)
        torch.zeros(1, device=self.device)  # make sure the HIP context exists before the library touches it
        h = C.c_void_p()
        rc = self.lib.relax_create(self.device.index, C.byref(h))
        if rc != 0:
            raise RuntimeError(f"relax_create failed ({rc}): {self.lib.relax_last_error(None).decode()}")
        self.h = h
        self.vit_dim = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.relax_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.relax_last_error(self.h).decode()}")

    # ---- weights ------------------------------------------------------------------------------
    def _marshal_state_dict(self, sd):
        names, arrays = [], []
        for k, v in sd.items():
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            v = np.asarray(v)
            if v.dtype.kind != "f":
                continue  # num_batches_tracked etc.
            names.append(k.encode())
            arrays.append(np.ascontiguousarray(v, dtype=np.float32))
        n = len(names)
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrays])
        cnames = (C.c_char_p * n)(*names)
        numels = (C.c_int64 * n)(*[a.size for a in arrays])
        return ptrs, cnames, numels, n, arrays

    def load_resnet50(self, state_dict):
        """state_dict: torchvision resnet50 key names -> fp32 arrays/tensors."""
        ptrs, names, numels, n, keep = self._marshal_state_dict(state_dict)
        self._check(self.lib.relax_load_resnet50(self.h, ptrs, names, numels, n), "relax_load_resnet50")
        del keep

    def load_vit(self, state_dict, name_model="vit_base"):
        """state_dict: DINO ViT key names -> fp32 arrays/tensors."""
        dim, depth, heads = VIT_CONFIGS[name_model]
        ptrs, names, numels, n, keep = self._marshal_state_dict(state_dict)
        self._check(self.lib.relax_load_vit(self.h, ptrs, names, numels, n, dim, depth, heads), "relax_load_vit")
        self.vit_dim = dim
        del keep

    def load_mlp_head(self, state_dict, scaler_scale, scaler_min, imputer_statistics=None):
        """state_dict: the reference Mlp's keys (src/model_regression.py:37-58); scaler_* / imputer_statistics: the
        MinMaxScaler.scale_/.min_ and SimpleImputer.statistics_ arrays of the reference's model/scaler/*.pkl."""
        ptrs, names, numels, n, keep = self._marshal_state_dict(state_dict)
        sc = np.ascontiguousarray(scaler_scale, dtype=np.float64)
        mn = np.ascontiguousarray(scaler_min, dtype=np.float64)
        st = None if imputer_statistics is None else np.ascontiguousarray(imputer_statistics, dtype=np.float64)
        rc = self.lib.relax_load_mlp_head(self.h, ptrs, names, numels, n, C.c_void_p(st.ctypes.data) if st is not None else None,
                                          C.c_void_p(sc.ctypes.data), C.c_void_p(mn.ctypes.data), int(sc.size))
        self._check(rc, "relax_load_mlp_head")
        del keep

    def mlp_head(self, features):
        """features fp32 [n, F] on the device -> predicted scores fp32 [n] (src/demo_test.py:177-208)."""
        features = features.to(self.device, torch.float32).contiguous()
        out = torch.empty((features.shape[0],), dtype=torch.float32, device=self.device)
        self._check(self.lib.relax_mlp_head(self.h, _ptr(features), features.shape[0], _ptr(out), _stream()), "relax_mlp_head")
        return out

    PRECISIONS = {"fp32": 0, "bf16x3": 1, "bf16x6": 2, "f16x2": 3}

    def set_precision(self, mode):
        """'fp32': exact fp32 products on the fp32 MFMA.  'bf16x6' (fp32-grade): each fp32 operand is held as three bf16
        values (hi + mid + lo, exact) and a*b = the six partial products of weight >= 2^-16 on the bf16 MFMA with fp32
        accumulation - as close to the exact result as the fp32 FMA chain at 6/16 of its matrix cycles.  'f16x2' (fp32-grade): each
        fp32 operand as two fp16 values of a power-of-two multiple of itself (22 bits; scales from bounds that hold for every input,
        csrc/h2.h), three partial products (four below K = 256) on the fp16 MFMA - the whole ViT-B (GEMMs and attention), ResNet-50's
        stem, 3x3 convolutions and layer3 / layer4; the launches without an f16x2 kernel run bf16x6 under it.  'bf16x3' (opt-in, lower precision): two bf16 values, three products, ~1e-5 norm-relative
        (the parity bar is 1e-3)."""
        self.set_option("gemm_precision", self.PRECISIONS[mode])

    def precision(self):
        """The arithmetic the engine's contraction kernel is in right now, read back from the library."""
        return {v: k for k, v in self.PRECISIONS.items()}[self.get_option("gemm_precision")]

    def set_option(self, key, value):
        self._check(self.lib.relax_set_option(self.h, key.encode(), int(value)), "relax_set_option")

    def get_option(self, key):
        v = C.c_int()
        self._check(self.lib.relax_get_option(self.h, key.encode(), C.byref(v)), "relax_get_option")
        return v.value

    def reserve(self, max_images):
        self._check(self.lib.relax_reserve(self.h, int(max_images)), "relax_reserve")

    # ---- stage A ------------------------------------------------------------------------------
    def _dev_u8(self, a):
        if isinstance(a, np.ndarray):
            a = torch.from_numpy(np.ascontiguousarray(a))
        a = a.to(self.device, non_blocking=True)
        if a.dtype != torch.uint8:
            raise TypeError(f"expected uint8, got {a.dtype}")
        return a.contiguous()

    def fragment_pairs(self, frames, top_n=TOP_N, want_scores=False, out_ori=None, out_diff=None):
        """frames: uint8 [T,2,H,W,3] BGR (frames[t,0] = sampled frame, frames[t,1] = the next one).
        -> dict(positions int32 [T,196,2], counts int32 [T], ori_frag, diff_frag uint8 [T,224,224,3][, scores])
        out_ori / out_diff: optional preallocated [T,224,224,3] uint8 views (slices of a batch buffer) to write into."""
        frames = self._dev_u8(frames)
        if frames.dim() != 5 or frames.shape[1] != 2 or frames.shape[4] != 3:
            raise ValueError(f"frames must be [T,2,H,W,3], got {tuple(frames.shape)}")
        T, _, H, W, _ = frames.shape
        dev = self.device
        positions = torch.empty((T, TOP_N, 2), dtype=torch.int32, device=dev)
        counts = torch.empty((T,), dtype=torch.int32, device=dev)
        ori = out_ori if out_ori is not None else torch.empty((T, TARGET, TARGET, 3), dtype=torch.uint8, device=dev)
        diff = out_diff if out_diff is not None else torch.empty((T, TARGET, TARGET, 3), dtype=torch.uint8, device=dev)
        for o in (ori, diff):
            if o.dtype != torch.uint8 or tuple(o.shape) != (T, TARGET, TARGET, 3) or not o.is_contiguous():
                raise ValueError("fragment_pairs: output buffers must be contiguous uint8 [T,224,224,3]")
        scores = torch.empty((T, (H // 16) * (W // 16)), dtype=torch.int32, device=dev) if want_scores else None
        frame_bytes = H * W * 3
        base = frames.data_ptr()
        rc = self.lib.relax_fragment_pairs(self.h, C.c_void_p(base), C.c_void_p(base + frame_bytes), 2 * frame_bytes,
                                           T, H, W, int(top_n), _ptr(positions), _ptr(counts), _ptr(ori), _ptr(diff),
                                           _ptr(scores), _stream())
        self._check(rc, "relax_fragment_pairs")
        out = dict(positions=positions, counts=counts, ori_frag=ori, diff_frag=diff)
        if want_scores:
            out["scores"] = scores.view(T, H // 16, W // 16)
        return out

    def fragment_image(self, images, top_n=TOP_N, want_scores=False, out=None):
        """images: uint8 [T,H,W,3] residual images (e.g. flow_to_rgb output). -> dict(positions, counts, frag[, scores])"""
        images = self._dev_u8(images)
        if images.dim() != 4 or images.shape[3] != 3:
            raise ValueError(f"images must be [T,H,W,3], got {tuple(images.shape)}")
        T, H, W, _ = images.shape
        dev = self.device
        positions = torch.empty((T, TOP_N, 2), dtype=torch.int32, device=dev)
        counts = torch.empty((T,), dtype=torch.int32, device=dev)
        frag = out if out is not None else torch.empty((T, TARGET, TARGET, 3), dtype=torch.uint8, device=dev)
        scores = torch.empty((T, (H // 16) * (W // 16)), dtype=torch.int32, device=dev) if want_scores else None
        rc = self.lib.relax_fragment_image(self.h, _ptr(images), H * W * 3, T, H, W, int(top_n), _ptr(positions),
                                           _ptr(counts), _ptr(frag), _ptr(scores), _stream())
        self._check(rc, "relax_fragment_image")
        out = dict(positions=positions, counts=counts, frag=frag)
        if want_scores:
            out["scores"] = scores.view(T, H // 16, W // 16)
        return out

    def gather_patches(self, images, positions, counts):
        images = self._dev_u8(images)
        T, H, W, _ = images.shape
        positions = positions.to(self.device, torch.int32).contiguous()
        counts = counts.to(self.device, torch.int32).contiguous()
        frag = torch.empty((T, TARGET, TARGET, 3), dtype=torch.uint8, device=self.device)
        rc = self.lib.relax_gather_patches(self.h, _ptr(images), H * W * 3, T, H, W, _ptr(positions), _ptr(counts),
                                           _ptr(frag), _stream())
        self._check(rc, "relax_gather_patches")
        return frag

    def merge_fragments(self, a, b, out=None):
        a, b = self._dev_u8(a), self._dev_u8(b)
        if a.shape != b.shape:
            raise ValueError("merge_fragments: shape mismatch")
        if out is None:
            out = torch.empty_like(a)
        self._check(self.lib.relax_merge_fragments(self.h, _ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()),
                    "relax_merge_fragments")
        return out

    def optical_flow(self, frames, want_flow=False, want_image=True):
        """frames uint8 [T,2,H,W,3] BGR -> (flow fp32 [T,H,W,2] | None, flow image uint8 [T,H,W,3] | None): Farneback flow
        with the reference's parameters and its flow_to_rgb visualisation (src/main_fragment_layerstack.py:313-316)."""
        frames = self._dev_u8(frames)
        if frames.dim() != 5 or frames.shape[1] != 2 or frames.shape[4] != 3:
            raise ValueError(f"frames must be [T,2,H,W,3], got {tuple(frames.shape)}")
        T, _, H, W, _ = frames.shape
        fl = torch.empty((T, H, W, 2), dtype=torch.float32, device=self.device) if want_flow else None
        im = torch.empty((T, H, W, 3), dtype=torch.uint8, device=self.device) if want_image else None
        fb = H * W * 3
        base = frames.data_ptr()
        rc = self.lib.relax_optical_flow(self.h, C.c_void_p(base), C.c_void_p(base + fb), 2 * fb, T, H, W, _ptr(fl), _ptr(im),
                                         _stream())
        self._check(rc, "relax_optical_flow")
        return fl, im

    def flow_to_rgb(self, flow):
        """flow fp32 [T,H,W,2] -> uint8 [T,H,W,3] (src/main_fragment_layerstack.py:162-175)."""
        flow = flow.to(self.device, torch.float32).contiguous()
        T, H, W, _ = flow.shape
        out = torch.empty((T, H, W, 3), dtype=torch.uint8, device=self.device)
        self._check(self.lib.relax_flow_to_rgb(self.h, _ptr(flow), T, H, W, _ptr(out), _stream()), "relax_flow_to_rgb")
        return out

    def resize_frames(self, frames, bilinear=True, lanczos=True, out_bilinear=None, out_lanczos=None):
        """frames uint8 [N,H,W,3] -> (bilinear, lanczos) uint8 [N,224,224,3] each (None if not requested), bit-identical
        to PIL's Image.resize((224,224), BILINEAR / LANCZOS) (the reference's whole-frame inputs)."""
        if isinstance(frames, np.ndarray):
            frames = torch.from_numpy(np.ascontiguousarray(frames))
        frames = frames.to(self.device, non_blocking=True)
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3:
            raise ValueError(f"frames must be uint8 [N,H,W,3], got {frames.dtype} {tuple(frames.shape)}")
        N, H, W, _ = frames.shape
        if frames.stride()[1:] != (W * 3, 3, 1):      # items may be strided (e.g. clip[:, 0]); pixels must be packed
            frames = frames.contiguous()
        item_stride = frames.stride(0) if N > 1 else H * W * 3
        ob = (out_bilinear if out_bilinear is not None else
              torch.empty((N, TARGET, TARGET, 3), dtype=torch.uint8, device=self.device)) if bilinear else None
        ol = (out_lanczos if out_lanczos is not None else
              torch.empty((N, TARGET, TARGET, 3), dtype=torch.uint8, device=self.device)) if lanczos else None
        self._check(self.lib.relax_resize_frames(self.h, _ptr(frames), item_stride, N, H, W, _ptr(ob), _ptr(ol), _stream()),
                    "relax_resize_frames")
        return ob, ol

    # ---- stage B ------------------------------------------------------------------------------
    def _frags(self, frags):
        frags = self._dev_u8(frags)
        if frags.dim() == 3:
            frags = frags.unsqueeze(0)
        if tuple(frags.shape[1:]) != (TARGET, TARGET, 3):
            raise ValueError(f"fragments must be [N,224,224,3], got {tuple(frags.shape)}")
        return frags

    def resnet50_features(self, frags, layer_stack=True, pool=True, taps=None):
        """frags uint8 [N,224,224,3] BGR -> (layer_stack fp32 [N,13120] | None, pool fp32 [N,2051] | None[, taps]).
        taps: optional iterable of tap indices (0..14) whose full activations [N,C,H,W] are returned too."""
        frags = self._frags(frags)
        N = frags.shape[0]
        dev = self.device
        ls = torch.empty((N, LAYER_STACK_DIM), dtype=torch.float32, device=dev) if layer_stack else None
        pl = torch.empty((N, RN50_POOL_DIM), dtype=torch.float32, device=dev) if pool else None
        tap_out, tap_ptrs = {}, None
        if taps is not None:
            arr = (C.c_void_p * 15)()
            for t in taps:
                c, s = RN50_TAP_SHAPES[t]
                tap_out[t] = torch.empty((N, c, s, s), dtype=torch.float32, device=dev)
                arr[t] = tap_out[t].data_ptr()
            tap_ptrs = arr
        rc = self.lib.relax_resnet50_features(self.h, _ptr(frags), N, _ptr(ls), _ptr(pl), tap_ptrs, _stream())
        self._check(rc, "relax_resnet50_features")
        return (ls, pl, tap_out) if taps is not None else (ls, pl)

    def resnet50_clip_features(self, frags, n_layer_stack):
        """frags uint8 [N,224,224,3]: the first n_layer_stack images are original fragments (-> layer stack fp32
        [n_layer_stack,13120]), the others residual fragments (-> pool fp32 [N - n_layer_stack, 2051]); ONE forward
        (src/main_fragment_layerstack.py:327-328).  Same values as resnet50_features on the respective images."""
        frags = self._frags(frags)
        N = frags.shape[0]
        n_ls = int(n_layer_stack)
        if not 0 <= n_ls <= N:
            raise ValueError(f"n_layer_stack={n_ls} outside [0, {N}]")
        ls = torch.empty((n_ls, LAYER_STACK_DIM), dtype=torch.float32, device=self.device)
        pl = torch.empty((N - n_ls, RN50_POOL_DIM), dtype=torch.float32, device=self.device)
        rc = self.lib.relax_resnet50_clip_features(self.h, _ptr(frags), N, n_ls, _ptr(ls) if n_ls else None, _ptr(pl) if N > n_ls else None,
                                                   _stream())
        self._check(rc, "relax_resnet50_clip_features")
        return ls, pl

    def vit_features(self, frags, tokens=False, pooled=True):
        """frags uint8 [N,224,224,3] BGR -> (tokens fp32 [N,196,dim] | None, pooled fp32 [N,3*dim] | None)"""
        if self.vit_dim is None:
            raise RuntimeError("load_vit first")
        frags = self._frags(frags)
        N = frags.shape[0]
        dev = self.device
        tk = torch.empty((N, 196, self.vit_dim), dtype=torch.float32, device=dev) if tokens else None
        pl = torch.empty((N, 3 * self.vit_dim), dtype=torch.float32, device=dev) if pooled else None
        self._check(self.lib.relax_vit_features(self.h, _ptr(frags), N, _ptr(tk), _ptr(pl), _stream()),
                    "relax_vit_features")
        return tk, pl

    # ---- whole clip ---------------------------------------------------------------------------
    def extract_clip(self, frames, resnet=True, vit=True, flow_images=None, flow=False):
        """frames uint8 [T,2,H,W,3] on the device -> per-frame features (all fp32, on the device):
             resnet: [T,15171] = layer-stack of the original fragment | pool of the residual fragment
             vit:    [T,4608]  = pool of the original fragment | pool of the residual fragment
        The residual fragment is the frame-difference fragment, merged 50/50 with the optical-flow
        fragment when flow_images (uint8 [T,H,W,3]) are supplied (src/main_fragment_layerstack.py:313-325)."""
        fr = self.fragment_pairs(frames)
        resid = fr["diff_frag"]
        if flow and flow_images is None:
            _, flow_images = self.optical_flow(frames)     # full ReLaX: Farneback + flow_to_rgb on the GPU
        if flow_images is not None:
            fl = self.fragment_image(flow_images)
            resid = self.merge_fragments(resid, fl["frag"])
        T = resid.shape[0]
        both = torch.cat([fr["ori_frag"], resid], dim=0)
        out = {"positions": fr["positions"], "counts": fr["counts"]}
        if resnet:
            ls, pool = self.resnet50_clip_features(both, T)
            out["resnet"] = torch.cat([ls, pool], dim=1)
        if vit:
            _, pooled = self.vit_features(both, tokens=False, pooled=True)
            out["vit"] = torch.cat([pooled[:T], pooled[T:]], dim=1)
        return out

    def _segment_means(self, out, blocks, counts):
        """out [len(counts), F] <- per-clip means; blocks: list of (src [rows, cols] fp32, first row, dst column)."""
        offs = np.ascontiguousarray(np.concatenate([[0], np.cumsum(counts)]), dtype=np.int32)   # host array, passed by value
        for src, row0, col0 in blocks:
            rc = self.lib.relax_segment_mean(self.h, _ptr(src), src.stride(0), src.shape[1], row0, C.c_void_p(offs.ctypes.data), len(counts),
                                             _ptr(out), out.stride(0), col0, _stream())
            self._check(rc, "relax_segment_mean")
        return out

    def clip_vectors(self, clips, resnet=True, vit=True, per_frame=False):
        """Several clips (list of uint8 [T,2,H,W,3] device tensors, any mix of resolutions) in ONE batched pass of
        both backbones -> fp32 [len(clips), F] per-clip mean vectors.  Bigger batches fill the 256 CUs better
        (more tiles per launch, fewer partial rounds).  A clip's row equals the row it gets alone to fp32 rounding; it is
        bit-identical across batch compositions (and therefore across ranks of a sharded run) only with
        set_option("gemm_split_k", 0): the default tail split cuts the last tiles of a GEMM along K by the batch size.
        per_frame=True: -> (matrix, [fp32 [T_i, F] per clip]) - the per-frame rows the reference saves per video
        (src/main_fragment_layerstack.py:345-354) next to their means."""
        counts = [int(c.shape[0]) for c in clips]
        n = sum(counts)
        both = torch.empty((2 * n, TARGET, TARGET, 3), dtype=torch.uint8, device=self.device)   # [originals | residuals]
        at = 0
        for c, t in zip(clips, counts):
            self.fragment_pairs(c, out_ori=both[at:at + t], out_diff=both[n + at:n + at + t])
            at += t
        F = (LAYER_STACK_DIM + RN50_POOL_DIM if resnet else 0) + (6 * self.vit_dim if vit else 0)
        out = torch.empty((len(clips), F), dtype=torch.float32, device=self.device)
        blocks, col = [], 0
        if resnet:
            ls, pool = self.resnet50_clip_features(both, n)        # layer stack of the originals, pool of the residuals
            blocks += [(ls, 0, 0), (pool, 0, LAYER_STACK_DIM)]
            col = LAYER_STACK_DIM + RN50_POOL_DIM
        if vit:
            _, pooled = self.vit_features(both, tokens=False, pooled=True)
            blocks += [(pooled, 0, col), (pooled, n, col + 3 * self.vit_dim)]
        self._segment_means(out, blocks, counts)
        if not per_frame:
            return out
        return out, self._per_frame_rows(blocks, counts)

    def rows_mean(self, rows):
        """fp32 [T, F] device tensor of per-frame rows -> their mean [F], by the same kernel (relax_segment_mean: the rows of a
        column added in order, one division) that reduces freshly computed rows: the dataset driver's resume path uses it so that
        a row rebuilt from the per-frame files is bit-identical to the one the first run returned."""
        rows = rows.to(device=self.device, dtype=torch.float32).contiguous()
        out = torch.empty((1, rows.shape[1]), dtype=torch.float32, device=self.device)
        return self._segment_means(out, [(rows, 0, 0)], [int(rows.shape[0])])[0]

    @staticmethod
    def _per_frame_rows(blocks, counts):
        """blocks as for _segment_means -> per clip the [T, F] matrix of its frames (file output only: aten copies)."""
        rows, at = [], 0
        order = sorted(blocks, key=lambda b: b[2])
        for t in counts:
            rows.append(torch.cat([src[row0 + at:row0 + at + t] for src, row0, _ in order], dim=1))
            at += t
        return rows

    def whole_frame_features(self, frames):
        """frames uint8 [N,H,W,3] BGR (whole sampled frames) -> (ResNet-50 layer-stack fp32 [N,13120], ViT pooled
        fp32 [N,2304]): the per-frame features of src/main_layer_stack.py:81-151 / src/demo_test.py:81-87, with the
        Pillow-exact resizes done on the GPU."""
        bil, lan = self.resize_frames(frames)
        ls, _ = self.resnet50_features(bil, layer_stack=True, pool=False)
        _, vp = self.vit_features(lan, tokens=False, pooled=True)
        return ls, vp

    def full_clip_vector(self, frames, flow_images=None, flow=False, whole_frames=None):
        """frames uint8 [T,2,H,W,3] -> fp32 [35203]: the vector src/demo_test.py:171-175 assembles
        (whole-frame ResNet-50 LS | whole-frame ViT | fragment ResNet-50 LS+pool | fragment ViT x2), each part averaged
        over the sampled frames.  Without flow_images the residual fragment is the frame-difference fragment alone.
        whole_frames uint8 [Ts,H,W,3]: all sampled frames when the last one has no pair (see full_clip_vectors)."""
        return self.full_clip_vectors([frames], flow=flow, flow_images=None if flow_images is None else [flow_images],
                                      whole_frames=None if whole_frames is None else [whole_frames])[0]

    def full_clip_vectors(self, clips, flow=True, flow_images=None, whole_frames=None):
        """Several clips -> fp32 [len(clips), 35203] in ONE batched pass of each backbone (3*T fragments / frames per
        clip: original fragment, residual fragment, whole frame).  Same layout as full_clip_vector.  Batch-invariant to
        fp32 rounding; bit for bit only with set_option("gemm_split_k", 0) (see clip_vectors).
        whole_frames: optional list of uint8 [Ts,H,W,3] per clip - ALL sampled frames (src/demo_test.py:76-87 averages the
        whole-frame features over every sampled frame, including a last one that has no `next` partner and therefore no
        pair); default: the first frame of every pair.
        = full_features(full_prepare(...)): full_prepare is the byte / HBM work (fragments, Farneback flow, resizes), full_features
        the contractions.  Running the two halves of consecutive batches on two streams was measured at 1.01 x
        (tools/flow_overlap_try.py) and is not done in the product; the split stays because the dataset driver stages on it."""
        return self.full_features(self.full_prepare(clips, flow=flow, flow_images=flow_images, whole_frames=whole_frames))

    def full_prepare(self, clips, flow=True, flow_images=None, whole_frames=None):
        """The input half of full_clip_vectors: fragments (difference + flow, merged), whole-frame resizes -> the two backbone
        input batches.  Everything is enqueued on the current stream."""
        counts = [int(c.shape[0]) for c in clips]
        wf = [c[:, 0] for c in clips] if whole_frames is None else list(whole_frames)
        wcounts = [int(w.shape[0]) for w in wf]
        n, nw = sum(counts), sum(wcounts)
        # ResNet batch: the layer-stack images first [ori | bilinear whole frames], then the pool images [residual]
        rn_in = torch.empty((2 * n + nw, TARGET, TARGET, 3), dtype=torch.uint8, device=self.device)
        vit_in = torch.empty((2 * n + nw, TARGET, TARGET, 3), dtype=torch.uint8, device=self.device)  # [ori | residual | lanczos]
        frag_bytes = TARGET * TARGET * 3
        at = aw = 0
        for i, (c, t) in enumerate(zip(clips, counts)):
            res = vit_in[n + at:n + at + t]
            self.fragment_pairs(c, out_ori=vit_in[at:at + t], out_diff=res)
            fimg = None if flow_images is None else flow_images[i]
            if flow and fimg is None:
                _, fimg = self.optical_flow(c)
            if fimg is not None:
                self.merge_fragments(res, self.fragment_image(fimg)["frag"], out=res)
            tw = wcounts[i]
            self.resize_frames(wf[i], out_bilinear=rn_in[n + aw:n + aw + tw], out_lanczos=vit_in[2 * n + aw:2 * n + aw + tw])
            at += t
            aw += tw
        # the fragments are the same for both backbones
        self._check(self.lib.relax_copy_bytes(self.h, _ptr(vit_in), _ptr(rn_in), n * frag_bytes, _stream()), "relax_copy_bytes")
        self._check(self.lib.relax_copy_bytes(self.h, _ptr(vit_in[n:]), _ptr(rn_in[n + nw:]), n * frag_bytes, _stream()), "relax_copy_bytes")
        return dict(rn_in=rn_in, vit_in=vit_in, counts=counts, wcounts=wcounts, n=n, nw=nw)

    def full_features(self, prep):
        """The backbone half of full_clip_vectors: prep from full_prepare (of this or another engine on the same device) ->
        fp32 [clips, 35203]."""
        rn_in, vit_in, counts, wcounts, n, nw = (prep[k] for k in ("rn_in", "vit_in", "counts", "wcounts", "n", "nw"))
        ls, pool = self.resnet50_clip_features(rn_in, n + nw)
        _, vp = self.vit_features(vit_in, tokens=False, pooled=True)
        d = self.vit_dim
        out = torch.empty((len(counts), LAYER_STACK_DIM + 3 * d + LAYER_STACK_DIM + RN50_POOL_DIM + 6 * d), dtype=torch.float32,
                          device=self.device)
        c0 = LAYER_STACK_DIM + 3 * d
        self._segment_means(out, [(ls, n, 0), (vp, 2 * n, LAYER_STACK_DIM)], wcounts)
        return self._segment_means(out, [(ls, 0, c0), (pool, 0, c0 + LAYER_STACK_DIM),
                                         (vp, 0, c0 + LAYER_STACK_DIM + RN50_POOL_DIM),
                                         (vp, n, c0 + LAYER_STACK_DIM + RN50_POOL_DIM + 3 * d)], counts)

    def clip_vector(self, frames, **kw):
        """Per-clip mean over frames of the concatenated features (src/demo_test.py:171-175)."""
        f = self.extract_clip(frames, **kw)
        parts = [f[k] for k in ("resnet", "vit") if k in f]
        out = torch.empty((1, sum(p.shape[1] for p in parts)), dtype=torch.float32, device=self.device)
        blocks, col = [], 0
        for p in parts:
            blocks.append((p, 0, col))
            col += p.shape[1]
        return self._segment_means(out, blocks, [int(parts[0].shape[0])])[0]

    # ---- operator level (tests / benches) -------------------------------------------------------
    def op_gemm(self, A, W, bias=None, residual=None, act=0, out=None):
        M, K = A.shape
        N = W.shape[0]
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=self.device)
        self._check(self.lib.relax_op_gemm(self.h, _ptr(A), _ptr(W), _ptr(bias), _ptr(residual), _ptr(out), M, N, K,
                                           act, _stream()), "relax_op_gemm")
        return out

    def op_conv2d_nhwc(self, x, w_packed, bias, residual, Cout, KH, KW, stride, pad, act):
        Nimg, H, W, Cin = x.shape
        Ho = (H + 2 * pad - KH) // stride + 1
        Wo = (W + 2 * pad - KW) // stride + 1
        out = torch.empty((Nimg, Ho, Wo, Cout), dtype=torch.float32, device=self.device)
        rc = self.lib.relax_op_conv2d_nhwc(self.h, _ptr(x), _ptr(w_packed), _ptr(bias), _ptr(residual), _ptr(out),
                                           Nimg, H, W, Cin, Cout, KH, KW, stride, pad, act, _stream())
        self._check(rc, "relax_op_conv2d_nhwc")
        return out

    def op_layernorm(self, x, g, b, eps):
        rows, dim = x.shape
        y = torch.empty_like(x)
        self._check(self.lib.relax_op_layernorm(self.h, _ptr(x), _ptr(g), _ptr(b), _ptr(y), rows, dim, eps, _stream()),
                    "relax_op_layernorm")
        return y

    def op_attention(self, qkv, n_img, heads):
        out = torch.empty((qkv.shape[0], heads * 64), dtype=torch.float32, device=self.device)
        self._check(self.lib.relax_op_attention(self.h, _ptr(qkv), _ptr(out), n_img, heads, _stream()),
                    "relax_op_attention")
        return out

    def op_bn_relu_maxpool(self, x, scale, shift):
        Nimg, H, W, Cc = x.shape
        y = torch.empty((Nimg, H // 2, W // 2, Cc), dtype=torch.float32, device=self.device)
        self._check(self.lib.relax_op_bn_relu_maxpool(self.h, _ptr(x), _ptr(scale), _ptr(shift), _ptr(y), Nimg, H, W,
                                                      Cc, _stream()), "relax_op_bn_relu_maxpool")
        return y

    def op_gap(self, x):
        Nimg, HW, Cc = x.shape
        out = torch.empty((Nimg, Cc), dtype=torch.float32, device=self.device)
        self._check(self.lib.relax_op_gap(self.h, _ptr(x), _ptr(out), Nimg, HW, Cc, Cc, _stream()), "relax_op_gap")
        return out

    def op_token_stats(self, x):
        Nimg, T, dim = x.shape
        out = torch.empty((Nimg, 3 * dim), dtype=torch.float32, device=self.device)
        self._check(self.lib.relax_op_token_stats(self.h, _ptr(x), _ptr(out), Nimg, T, dim, _stream()),
                    "relax_op_token_stats")
        return out

    # ---- measurement ----------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._check(self.lib.relax_profile_enable(self.h, int(bool(on))), "relax_profile_enable")

    def profile_read(self, kind):
        ms, work, n = C.c_double(), C.c_double(), C.c_int64()
        self._check(self.lib.relax_profile_read(self.h, kind, C.byref(ms), C.byref(work), C.byref(n)),
                    "relax_profile_read")
        return ms.value, work.value, n.value


def pack_conv_weight(w_oihw, cin_pad=None):
    """OIHW fp32 -> [Cout, Kpad] with k = (dy*KW+dx)*Cin_pad + c, zero padded to a multiple of 32
    (the layout relax_op_conv2d_nhwc expects; the model loader does the same on the host in C++)."""
    w = np.asarray(w_oihw, dtype=np.float32)
    co, ci, kh, kw = w.shape
    cp = cin_pad or ci
    k = kh * kw * cp
    kpad = -(-k // 32) * 32
    out = np.zeros((co, kpad), dtype=np.float32)
    t = np.zeros((co, kh, kw, cp), dtype=np.float32)
    t[..., :ci] = w.transpose(0, 2, 3, 1)
    out[:, :k] = t.reshape(co, k)
    return out
