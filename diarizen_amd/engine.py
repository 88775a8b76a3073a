"""Python handle on the C-ABI engine (include/dzn.h).

PyTorch is plumbing here: it owns device buffers and the HIP stream; every forward is one
C call into libdzn_hip.so with raw pointers.  No CPU fallback exists.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Mapping, Optional

import numpy as np
import torch

from . import _lib
from ._lib import (DZN_F32, DZN_F64, DZN_I64, DZN_PREC_BF16, DZN_PREC_F32, DZN_PREC_F32_SPLIT, DznConfig,
                   check)
from .configs import EmbConfig, SegConfig

PRECISIONS = {"f32": DZN_PREC_F32, "fp32": DZN_PREC_F32, "bf16": DZN_PREC_BF16,
              "f32s": DZN_PREC_F32_SPLIT, "f32_split": DZN_PREC_F32_SPLIT, "f32h": _lib.DZN_PREC_F32_H2,
              "f16": _lib.DZN_PREC_F16, "fp16": _lib.DZN_PREC_F16}


def make_dzn_config(seg: SegConfig, emb: Optional[EmbConfig], max_batch: int, max_samples: int,
                    precision: str = "f32h") -> DznConfig:
    c = DznConfig()
    c.struct_size = C.sizeof(DznConfig)
    c.precision = PRECISIONS[precision]
    c.max_batch = int(max_batch)
    c.max_samples = int(max_samples)
    c.extractor_layer_norm = int(seg.extractor_layer_norm)
    c.normalize_waveform = int(seg.normalize_waveform)
    c.n_conv = len(seg.conv_channels)
    for i, (ch, k, s) in enumerate(zip(seg.conv_channels, seg.conv_kernels, seg.conv_strides)):
        c.conv_ch[i], c.conv_k[i], c.conv_s[i] = ch, k, s
    c.embed_dim = seg.embed_dim
    c.total_heads = seg.total_heads
    c.n_layers = seg.n_layers
    c.layer_norm_first = int(seg.layer_norm_first)
    c.pos_conv_kernel = seg.pos_conv_kernel
    c.pos_conv_groups = seg.pos_conv_groups
    c.num_buckets = seg.num_buckets
    c.max_distance = seg.max_distance
    for i in range(seg.n_layers):
        heads = seg.remaining_heads[i]
        c.use_attention[i] = int(len(heads) > 0)
        c.n_heads[i] = len(heads)
        for j, hd in enumerate(heads):
            c.head_idx[i][j] = hd
        c.use_ffn[i] = 1
        c.ffn_dim[i] = seg.ffn_dims[i]
    c.attention_in = seg.attention_in
    c.ffn_hidden = seg.ffn_hidden
    c.conf_heads = seg.conf_heads
    c.conf_layers = seg.conf_layers
    c.conf_kernel = seg.conf_kernel
    c.n_classes = seg.n_classes
    c.max_speakers_per_chunk = seg.max_speakers_per_chunk
    c.max_speakers_per_frame = seg.max_speakers_per_frame
    c.has_embedding = int(emb is not None)
    if emb is not None:
        c.embed_out_dim = emb.embed_dim
        c.num_mel_bins = emb.num_mel_bins
    return c


class Engine:
    """One engine per HIP device.  `seg_state` uses the reference Model's state_dict keys;
    `emb_state` the WeSpeakerResNet34 keys ("resnet.conv1.weight", ...)."""

    def __init__(self, seg: SegConfig, seg_state: Mapping[str, torch.Tensor],
                 emb: Optional[EmbConfig] = None,
                 emb_state: Optional[Mapping[str, torch.Tensor]] = None, *, max_batch: int = 32,
                 max_samples: int = 128000, precision: str = "f32h",
                 device: Optional[torch.device] = None):
        if not torch.cuda.is_available():
            raise _lib.DznError("no HIP device: diarizen_amd has no CPU path")
        self.lib = _lib.load()
        self.device = torch.device(device or "cuda:0")
        self.seg, self.emb = seg, emb
        self.max_batch, self.max_samples, self.precision = max_batch, max_samples, precision
        self._h = C.c_void_p()
        cfg = make_dzn_config(seg, emb if emb_state is not None else None, max_batch, max_samples,
                              precision)
        with torch.cuda.device(self.device):
            check(self.lib.dzn_create(C.byref(cfg), C.byref(self._h)), None, "dzn_create")
            self._load(seg_state, "")
            if emb_state is not None:
                self._load(emb_state, "embedding.")
            check(self.lib.dzn_finalize_weights(self._h), self._h, "dzn_finalize_weights")

    # ------------------------------------------------------------------ weights
    def _load(self, state: Mapping[str, torch.Tensor], prefix: str) -> None:
        for key, t in state.items():
            t = t.detach().cpu().contiguous()
            if t.dtype == torch.float32:
                dt = DZN_F32
            elif t.dtype == torch.float64:
                dt = DZN_F64
            elif t.dtype == torch.int64:
                dt = DZN_I64
            else:
                t = t.float()
                dt = DZN_F32
            shape = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            check(self.lib.dzn_load_tensor(self._h, (prefix + key).encode(), C.c_void_p(t.data_ptr()),
                                           shape, t.dim(), dt), self._h, f"dzn_load_tensor({key})")

    # ------------------------------------------------------------------ info
    def num_frames(self, num_samples: int) -> int:
        return self.lib.dzn_num_frames(self._h, num_samples)

    @property
    def workspace_bytes(self) -> int:
        return self.lib.dzn_workspace_bytes(self._h)

    @property
    def num_ignored_keys(self) -> int:
        return self.lib.dzn_num_ignored(self._h)

    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ forwards
    def segment(self, wave: torch.Tensor, want_logp: bool = True, want_multilabel: bool = True):
        """wave: f32 [B, N] on the device.  Returns (logp [B, L, n_classes] or None,
        multilabel u8 [B, L, S] or None) — enqueue only, caller synchronises."""
        assert wave.is_cuda and wave.dtype == torch.float32 and wave.dim() == 2 and wave.is_contiguous()
        B, N = wave.shape
        L = self.num_frames(N)
        logp = (torch.empty((B, L, self.seg.n_classes), device=wave.device, dtype=torch.float32)
                if want_logp else None)
        ml = (torch.empty((B, L, self.seg.max_speakers_per_chunk), device=wave.device, dtype=torch.uint8)
              if want_multilabel else None)
        check(self.lib.dzn_segment_forward(
            self._h, C.c_void_p(wave.data_ptr()), B, N,
            C.c_void_p(logp.data_ptr()) if logp is not None else None,
            C.c_void_p(ml.data_ptr()) if ml is not None else None, self._stream()),
            self._h, "dzn_segment_forward")
        return logp, ml

    def embed(self, wave: torch.Tensor, masks: torch.Tensor) -> torch.Tensor:
        """wave f32 [B, N], masks f32 [B, S, L] (device) -> embeddings f32 [B, S, dim]."""
        assert wave.is_cuda and wave.dtype == torch.float32 and wave.dim() == 2 and wave.is_contiguous()
        assert masks.is_cuda and masks.dtype == torch.float32 and masks.dim() == 3 and masks.is_contiguous()
        B, N = wave.shape
        _, S, L = masks.shape
        out = torch.empty((B, S, self.emb.embed_dim), device=wave.device, dtype=torch.float32)
        check(self.lib.dzn_embed_forward(self._h, C.c_void_p(wave.data_ptr()),
                                         C.c_void_p(masks.data_ptr()), B, S, N, L,
                                         C.c_void_p(out.data_ptr()), self._stream()),
              self._h, "dzn_embed_forward")
        return out

    def prepare_masks(self, multilabel: torch.Tensor, median_size: int = 11,
                      exclude_overlap: bool = True, min_num_frames: int = -1,
                      want_masks: bool = True):
        """multilabel u8 [B, L, S] (device) -> (filtered u8 [B, L, S], masks f32 [B, S, L])."""
        assert multilabel.is_cuda and multilabel.dtype == torch.uint8 and multilabel.is_contiguous()
        B, L, S = multilabel.shape
        filt = torch.empty_like(multilabel)
        masks = (torch.empty((B, S, L), device=multilabel.device, dtype=torch.float32)
                 if want_masks else None)
        check(self.lib.dzn_prepare_masks(self._h, C.c_void_p(multilabel.data_ptr()), B, L,
                                         int(median_size), int(exclude_overlap), int(min_num_frames),
                                         C.c_void_p(filt.data_ptr()),
                                         C.c_void_p(masks.data_ptr()) if masks is not None else None,
                                         self._stream()), self._h, "dzn_prepare_masks")
        return filt, masks

    def debug_fetch(self, name: str) -> np.ndarray:
        n = C.c_int64(0)
        check(self.lib.dzn_debug_fetch(self._h, name.encode(), None, 0, C.byref(n)), self._h,
              f"dzn_debug_fetch({name})")
        out = np.empty(n.value, dtype=np.float32)
        check(self.lib.dzn_debug_fetch(self._h, name.encode(), C.c_void_p(out.ctypes.data), n.value,
                                       C.byref(n)), self._h, f"dzn_debug_fetch({name})")
        return out

    def embed_skip_stats(self):
        """(windows seen by embed(), windows whose ResNet trunk pass was skipped because no speaker was active)"""
        w, k = C.c_int64(0), C.c_int64(0)
        check(self.lib.dzn_embed_skip_stats(self._h, C.byref(w), C.byref(k)), self._h, "dzn_embed_skip_stats")
        return w.value, k.value

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.dzn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass
