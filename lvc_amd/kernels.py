"""Python host bindings of the gfx950 kernels (one function per C-ABI entry point of
include/lvc_amd.h).  torch is used for device memory and streams only; all arithmetic of the
hot path happens inside liblvc_amd.so.  Every function requires CUDA(HIP) tensors and raises if
the native library is missing -- there is deliberately no CPU path here.
"""
import ctypes

import torch

from . import _lib
from ._lib import LvcNativeError, c_double, c_float, c_int, c_longlong, c_void_p, check, ptr

BK = 32  # gemm-K chunk of the implicit-GEMM kernel
BN = 128


def _stream(t):
    return c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _req_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("lvc_amd kernels need device tensors (got a CPU tensor); there is no CPU path")


# --------------------------------------------------------------------------- convolution / GEMM
class PackedConv:
    """Weights of one conv/linear layer in the kernel's layout + folded per-channel affine.

    w_packed: [Kpad, Kg] fp32, rows = out channel (zero rows up to a multiple of 128),
              k = (c//32, r, s, c%32) (mode 0) or (r, 8 pixels x 4 ch) (mode 1, the 7x7 stem).
    scale/shift: y = conv * scale + shift  (FrozenBN fold and/or bias), or None.
    """

    __slots__ = ("w", "scale", "shift", "K", "C", "R", "S", "stride", "pad", "Kg", "mode", "_w3", "_w2h", "_w2s", "two_acc",
                 "slot", "state", "_last_one", "__weakref__")

    def __init__(self, w, scale, shift, K, C, R, S, stride, pad, Kg, mode):
        self.w, self.scale, self.shift = w, scale, shift
        self.K, self.C, self.R, self.S, self.stride, self.pad, self.Kg, self.mode = K, C, R, S, stride, pad, Kg, mode
        self._w3 = None
        self._w2h = None
        self._w2s = None
        self.two_acc = False   # True: never the single-accumulator form of the 3x3 fp16-split kernel (see HALO_S1)
        # range routing of THIS layer (`check_conv_error_word`): its own range word in the workspace, and the tier it runs on --
        # 0: as configured (one accumulator where the policy allows it, |a| <= 4094), 1: two accumulators (|a| <= 65504),
        # 2: the range-free bf16x3 kernels.  `state` may be a dict the owning module keeps across re-packs.
        self.slot = _new_range_slot(self)
        self.state = {"tier": 0}
        self._last_one = False

    @property
    def last_one(self):
        """True when the layer's most recent launch ran on the single-accumulator form (|a| <= 4094)."""
        return self._last_one

    @last_one.setter
    def last_one(self, one):
        # ... and, per (device, stream) workspace, the (tier, form) that launch ran on: a range word is read with the workspace of the
        # stream it was raised on, possibly after another stream's pass already moved the layer (PipelinedInference) -- the reaction
        # is chosen from the form the word was raised by, not from the layer's present one
        self._last_one = bool(one)
        dev = self.w.device
        if dev.type == "cuda":
            # one layer may launch several forms in a pass (the RPN head: single-accumulator Winograd on p2 / p3, two accumulators on
            # p4 - p6, all on this slot): the narrowest form seen at this tier decides the reaction, not the last launch
            at = self.state.setdefault("at", {})
            key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
            prev = at.get(key)
            at[key] = (self.state["tier"], bool(one) or (prev is not None and prev[0] == self.state["tier"] and prev[1]))

    def _split(self, planes):
        n = self.w.numel()
        dt = torch.bfloat16 if planes == 3 else torch.float16
        out = torch.empty((planes,) + tuple(self.w.shape), device=self.w.device, dtype=dt)
        err = _conv_error_view(self.w.device) if planes == 2 else None
        check(_lib.lib().lvc_split_weights(ptr(self.w), c_longlong(n), c_int(planes), ptr(out), ptr(err), _stream(self.w)),
              "lvc_split_weights")
        return out

    def split3(self):
        """[3, Kpad, Kg] bf16 planes (hi, mid, lo) of the packed weights: w == hi + mid + lo exactly."""
        if self._w3 is None:
            self._w3 = self._split(3)
        return self._w3

    def split2h(self):
        """[2, Kpad, Kg] fp16 planes of the packed weights for the two-way fp16 split (csrc/conv3x3_halo_h2.hip):
        w1 = fp16(w), w2 = fp16((w - w1) * 2048)  ->  w ~= w1 + w2 / 2048 to 2^-23 relative (|w| >= 2.4e-4).  A weight
        beyond the fp16 range raises bit 1 of the conv error word (`check_conv_error_word`: use LVC_CONV_SPLIT=bf16x3)."""
        if self._w2h is None:
            self._w2h = self._split(2)
        return self._w2h


    def split2s(self):
        """([2, Kpad, Kg] fp16 planes of the ROW-SCALED weights, scale' [K] fp32) for the single-accumulator form of the two-way
        fp16 split (csrc/conv3x3_halo_s1.hip): row k is multiplied by 2^e_k (largest entry into [2^13, 2^14)) and split
        without a plane scale, w 2^e = w1 + w2; scale' = (the layer's per-channel scale or 1) * 2^-(e_k + 4) undoes that and
        the kernel's 2^4 activation scale in the epilogue.  Powers of two: the scaling itself is exact."""
        if self._w2s is None:
            rows, Kg = self.w.shape
            planes = torch.empty((2, rows, Kg), device=self.w.device, dtype=torch.float16)
            fac = torch.empty(rows, device=self.w.device, dtype=torch.float32)
            check(_lib.lib().lvc_split_weights_rowscaled(ptr(self.w), c_int(rows), c_int(Kg), ptr(planes), ptr(fac), _stream(self.w)),
                  "lvc_split_weights_rowscaled")
            fac = fac[: self.K]
            self._w2s = (planes, (fac * self.scale if self.scale is not None else fac).contiguous())
        return self._w2s


_RANGE_SLOTS = 1024
_NEXT_SLOT = [0]
_SLOT_OWNERS = {}
RANGE_EPOCH = 0      # counts the re-routings: passes launched under an older epoch ran on the narrower kernels


def _new_range_slot(owner):
    """Slots 1 .. 1023, handed out round-robin; a slot shared by two layers (more than 1023 packed layers alive) only makes
    the re-routing coarser (both move to the wider tier)."""
    import weakref

    _NEXT_SLOT[0] = _NEXT_SLOT[0] % (_RANGE_SLOTS - 1) + 1
    slot = _NEXT_SLOT[0]
    owners = [r for r in _SLOT_OWNERS.get(slot, []) if r() is not None]
    owners.append(weakref.ref(owner))
    _SLOT_OWNERS[slot] = owners
    return slot


def conv_affine(bias=None, bn=None, eps=1e-5):
    """(scale, shift) of y = conv * scale + shift: FrozenBatchNorm2d fold (reference detectron2/layers/batch_norm.py:
    45-65: scale = w * rsqrt(var + eps), shift = b - mean * scale) and/or the conv bias; None where absent."""
    scale = shift = None
    if bn is not None:
        bw, bb, rm, rv = [t.detach().float() for t in bn]
        scale = bw * (rv + eps).rsqrt()
        shift = bb - rm * scale
        if bias is not None:
            shift = shift + bias.detach().float() * scale
    elif bias is not None:
        shift = bias.detach().float().clone()
    if scale is not None:
        scale = scale.contiguous()
    if shift is not None:
        shift = shift.contiguous()
    return scale, shift


def _pack_weights(weight, scale, rows_pad, cin_pad, mode, planes=0):
    """Packed fp32 operand and, with planes = 2 / 3, its fp16 / bf16 split planes from the same launch (else None)."""
    K, C, R, S = weight.shape
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    wp = torch.empty(rows_pad, R * S * cin_pad, device=w.device, dtype=torch.float32)
    if planes:
        out = torch.empty((planes,) + tuple(wp.shape), device=w.device, dtype=torch.float16 if planes == 2 else torch.bfloat16)
        err = _conv_error_view(w.device) if planes == 2 else None
        check(_lib.lib().lvc_pack_split_conv_weights(ptr(w), ptr(scale), ptr(wp), ptr(out), c_int(planes), ptr(err), c_int(K),
                                                     c_int(C), c_int(R), c_int(S), c_int(rows_pad), c_int(cin_pad),
                                                     c_int(mode), _stream(w)), "lvc_pack_split_conv_weights")
        return wp, out
    check(_lib.lib().lvc_pack_conv_weights(ptr(w), ptr(scale), ptr(wp), c_int(K), c_int(C), c_int(R), c_int(S),
                                           c_int(rows_pad), c_int(cin_pad), c_int(mode), _stream(w)),
          "lvc_pack_conv_weights")
    return wp, None


def _planes_hint(R, S, contraction, split):
    """The split the conv routing will most likely ask of a layer (conv2d_nhwc: two-way fp16 for the 3x3 layers and the
    1x1 layers with a long contraction, three-way bf16 otherwise); a wrong guess only costs the lazy split later."""
    if CONV_ENGINE != "bf16x3":
        return 0
    if (split or CONV_SPLIT) == "f16x2" and ((R == 3 and S == 3) or (R == 1 and S == 1 and contraction >= _H2_PW_MIN_C)):
        return 2
    return 3


def _with_planes(pc, planes, out):
    if planes == 2:
        pc._w2h = out
    elif planes == 3:
        pc._w3 = out
    return pc


def pack_conv(weight, bias=None, bn=None, stride=1, pad=0, eps=1e-5, stem=False, affine=None, split=None):
    """weight: [K, C, R, S] (OIHW, the reference's state_dict layout) on the target device.
    bn: None or (weight, bias, running_mean, running_var) of a FrozenBatchNorm2d; affine: a precomputed
    `conv_affine(bias, bn, eps)` (the fold only changes when those tensors do, the weights change every step).
    stem=True packs the 3-channel 7x7 stem for the NHWC4 "row mode" of the kernel.
    """
    _req_cuda(weight)
    K, C, R, S = weight.shape
    dev = weight.device
    Kpad = (K + BN - 1) // BN * BN
    if stem:
        assert C <= 4 and S <= 8
        w = weight.detach().float()
        wk = torch.zeros(K, R, 8, 4, device=dev, dtype=torch.float32)
        wk[:, :, :S, :C] = w.permute(0, 2, 3, 1)
        Kg = R * BK
        wp = torch.zeros(Kpad, Kg, device=dev, dtype=torch.float32)
        wp[:K] = wk.reshape(K, Kg)
        Cphys, mode = 4, 1
    else:
        assert C % BK == 0, "implicit-GEMM kernel needs in_channels % 32 == 0 (got {})".format(C)
        Kg = R * S * C
        # k = (c // 32, r, s, c % 32): the taps of one 32-channel chunk are consecutive gemm-k chunks
        planes = _planes_hint(R, S, C, split)
        wp, pl = _pack_weights(weight, None, Kpad, C, 0, planes)
        scale, shift = affine if affine is not None else conv_affine(bias, bn, eps)
        return _with_planes(PackedConv(wp, scale, shift, K, C, R, S, stride, pad, Kg, 0), planes, pl)
    scale, shift = affine if affine is not None else conv_affine(bias, bn, eps)
    return PackedConv(wp, scale, shift, K, Cphys, R, S, stride, pad, Kg, mode)


PREPACK = __import__("os").environ.get("LVC_PREPACK", "1") != "0"


class PrepackPlan:
    """The packed operands of many layers in a few launches (csrc/weights.hip pack_group_kernel) instead of two or three launches
    per layer: what `pack_conv` (+ `PackedConv.split2s`) and `pack_conv_dgrad` produce, for the layers whose parameters an
    optimizer step just changed.  jobs: list of dicts {"weight", "kind": "fwd" | "dgrad", "stride", "pad", "affine": (scale, shift),
    "two_acc", "tier", "scale"}; `.packed`: the PackedConv objects (same fields as the per-layer functions fill).
    The plan owns its buffers: `relaunch()` packs the (changed) parameters into the SAME operands again -- the host side of a
    training step's re-packing is then one kernel-argument table per 24 layers, not ~10 tensor views per layer."""

    def __init__(self, jobs):
        import ctypes

        dev = jobs[0]["weight"].device
        metas, total = [], 0

        def take(nbytes):
            nonlocal total
            o = total
            total += (nbytes + 255) // 256 * 256
            return o

        for jb in jobs:
            w = jb["weight"]
            Kc, C, R, S = w.shape
            if jb["kind"] == "fwd":
                assert C % BK == 0
                rows_pad, cin_pad, mode = (Kc + BN - 1) // BN * BN, C, 0
                hint = _planes_hint(R, S, C, None)
                one = (hint == 2 and not jb.get("two_acc") and jb.get("tier", 0) == 0
                       and ((R == 3 and HALO_S1 == 2) or (R == 1 and PW_S1 == 2 and C >= _PW_S1_ONE_MIN_C)))
                fmt = 4 if one else hint
            else:
                cin_pad = (Kc + 31) // 32 * 32
                rows_pad, mode = (C + BN - 1) // BN * BN, 1
                fmt = _planes_hint(R, S, cin_pad, DGRAD_SPLIT)
            Kg = R * S * cin_pad
            nplanes = 3 if fmt == 3 else 2 if fmt in (2, 4) else 0
            metas.append((rows_pad, cin_pad, mode, fmt, Kg, take(rows_pad * Kg * 4), take(nplanes * rows_pad * Kg * 2) if nplanes else -1,
                          take(rows_pad * 4) if fmt == 4 else -1))
        self.buf = buf = torch.empty(total, dtype=torch.uint8, device=dev)
        base = buf.data_ptr()
        self.n = n = len(jobs)
        self.ptrs = ptrs = (ctypes.c_void_p * (6 * n))()
        self.shapes = shapes = (c_int * (8 * n))()
        self.packed, self.planes, self.keep = [], [], []
        self.reusable = True
        for j, (jb, (rows_pad, cin_pad, mode, fmt, Kg, o_wp, o_pl, o_fac)) in enumerate(zip(jobs, metas)):
            w = jb["weight"].detach()
            if w.dtype != torch.float32 or not w.is_contiguous():
                w = w.float().contiguous()
                self.reusable = False          # a converted copy: its address is not the parameter's
            self.keep.append(w)
            Kc, C, R, S = w.shape
            aff = jb.get("affine") or (None, None)
            wp = buf[o_wp: o_wp + rows_pad * Kg * 4].view(torch.float32).view(rows_pad, Kg)
            pl = fac = None
            if fmt in (2, 4):
                pl = buf[o_pl: o_pl + 2 * rows_pad * Kg * 2].view(torch.float16).view(2, rows_pad, Kg)
            elif fmt == 3:
                pl = buf[o_pl: o_pl + 3 * rows_pad * Kg * 2].view(torch.bfloat16).view(3, rows_pad, Kg)
            if fmt == 4:
                fac = buf[o_fac: o_fac + rows_pad * 4].view(torch.float32)
            dg_scale = jb.get("scale") if mode == 1 else None
            self.keep.append((dg_scale, aff))
            ptrs[6 * j: 6 * j + 6] = [w.data_ptr(), dg_scale.data_ptr() if dg_scale is not None else None,
                                      aff[0].data_ptr() if (fmt == 4 and aff[0] is not None) else None, base + o_wp,
                                      (base + o_pl) if pl is not None else None, (base + o_fac) if fac is not None else None]
            shapes[8 * j: 8 * j + 8] = [Kc, C, R, S, rows_pad, cin_pad, mode, fmt]
            if mode == 0:
                pc = PackedConv(wp, aff[0], aff[1], Kc, C, R, S, jb["stride"], jb["pad"], Kg, 0)
            else:
                pc = PackedConv(wp, None, None, C, cin_pad, R, S, 1, R - 1 - jb["pad"], Kg, 0)
            self.packed.append(pc)
            self.planes.append((fmt, pl, fac[:Kc] if fac is not None else None))
        self._err = _conv_error_view(dev)
        self.relaunch()

    def relaunch(self):
        for pc, (fmt, pl, fac) in zip(self.packed, self.planes):
            pc._w2h = pl if fmt == 2 else None          # lazily made planes of the previous parameters go too
            pc._w3 = pl if fmt == 3 else None
            pc._w2s = (pl, fac) if fmt == 4 else None
            pc.state.pop("_wino", None)
        check(_lib.lib().lvc_pack_group(c_int(self.n), self.ptrs, self.shapes, ptr(self._err), _stream(self.buf)), "lvc_pack_group")


def prepack_group(jobs):
    return PrepackPlan(jobs).packed if jobs else []


def pack_linear(weight, bias=None, split=None, two_acc=True):
    """weight [K_out, K_in] -> a 1x1 'conv' over M x 1 x 1 x K_in rows.  split: the operand split the GEMM will be
    asked for (None = LVC_CONV_SPLIT), so that the matching planes are produced by the packing launch.
    two_acc (default True): fully-connected layers feed scores, box deltas and kNN rankings -- they keep the main + cross
    accumulator form of the fp16 split (kernels.PW_S1); the descriptor network's linears pass False."""
    pc = pack_conv(weight[:, :, None, None], bias=bias, split=split)
    pc.two_acc = bool(two_acc)
    return pc


_CONV_WS = {}


def conv_workspace(device):
    """Per-(device, stream) scratch of the stream-K conv kernel (zeroed once; see include/lvc_amd.h)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _CONV_WS.get(key)
    if ws is None:
        lib = _lib.lib()
        lib.lvc_conv_workspace_bytes.restype = c_longlong
        ws = torch.zeros(lib.lvc_conv_workspace_bytes(), dtype=torch.uint8, device=device)
        _CONV_WS[key] = ws
    return ws


_ERR_OFF = 1024 * 256 * 128 * 4 + 1024 * 4      # byte offset of the range / error words: word 0 = shared, words 1.. = per layer


def _conv_error_view(device):
    ws = conv_workspace(device)
    return ws[_ERR_OFF: _ERR_OFF + 4]


def _range_words(device):
    ws = conv_workspace(device)
    return ws[_ERR_OFF: _ERR_OFF + 4 * _RANGE_SLOTS].view(torch.int32)


def range_summary(device):
    """[1] int32 device tensor, non-zero iff any error / range word of this stream's workspace is set: lets a caller fold the
    check into a device->host read it does anyway (`roi_heads.instances_from_batched`) and call `check_conv_error_word` only
    when something is up."""
    return _range_words(device).amax().view(1)


def conv_error_word(device):
    """OR of the kernels' error words (0 = fine; bit 0: spin timeout, bit 1: a finite operand beyond the range of the form, bit 2: a
    non-finite operand in a range-checked kernel); reading it synchronises."""
    e = 0
    for w in _range_words(device).tolist():
        e |= w
    return e


def clear_conv_error_word(device):
    _range_words(device).zero_()


class Fp16RangeError(LvcNativeError):
    """An operand beyond the range of a two-way fp16 split kernel reached it (|a| > 4094 on the single-accumulator form,
    > 65504 on the two-accumulator form, or NaN): the results of that pass are invalid.  `rerouted`: the layers that raised
    their own range word were moved to the next wider form (`check_conv_error_word`); the model entry points run the pass
    again.  Otherwise (shared word: stem, weights, descriptor network) they switch the process to the bf16x3 kernels."""

    rerouted = False


_RANGE_FALLBACK_LOGGED = False
_LOGGED_ONCE = set()


def _log_once(key, msg, *args):
    if key not in _LOGGED_ONCE:
        _LOGGED_ONCE.add(key)
        import logging

        logging.getLogger("lvc_amd").warning(msg, *args)


def use_range_free_split(reason=""):
    """Switch every conv/GEMM of this process from the two-way fp16 split to the exact three-way bf16 split (no range
    limit, ~1.5x slower); logged once.  Packed layers keep both plane sets lazily, so this takes effect at the next call.
    Only for conditions that cannot be pinned on one layer (the shared error word); a layer that overflows its own range
    word is re-routed alone (`check_conv_error_word`)."""
    global CONV_SPLIT, _RANGE_FALLBACK_LOGGED
    CONV_SPLIT = "bf16x3"
    if not _RANGE_FALLBACK_LOGGED:
        _RANGE_FALLBACK_LOGGED = True
        import logging

        logging.getLogger("lvc_amd").warning(
            "an operand beyond fp16's range (|a| > 65504 or NaN) reached an fp16x2 kernel that has no per-layer range word%s; "
            "re-running on the range-free bf16x3 kernels and keeping them for the rest of the process", (" (" + reason + ")") if reason else "")


def check_conv_error_word(device):
    """Raise for a set bit of the conv workspace's error words.  Bit 0 (any word): a stream-K worker timed out waiting for a
    partial tile.  Bit 1 of a LAYER's word: an activation left the range of the form that layer ran on -> the layer moves to
    the next wider one -- single accumulator (|a| <= 4094) -> two accumulators (|a| <= 65504) -> bf16x3 (no limit) -- and
    stays there; every other layer keeps its kernels.  Bit 1 of the shared word: `Fp16RangeError` without `rerouted`
    (process-wide bf16x3, `use_range_free_split`).  The words are cleared: every pass is judged on its own.
    Synchronises: call it where the results are read anyway."""
    global RANGE_EPOCH
    words = _range_words(device).tolist()
    if not any(words):
        return
    if any(w & 1 for w in words):
        raise LvcNativeError("conv/GEMM kernel: a stream-K worker timed out waiting for a partial tile")
    clear_conv_error_word(device)
    moved, stale = [], 0
    ws_key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    # bit 1: a finite operand beyond the form's range -- that layer moves.  bit 2 alone: the layer saw inf (NaN is not tracked); when
    # another layer reports bit 1 in the same pass that is what the overflowing layer handed down, and the layer stays where it is
    # (the repeated pass will tell); with no bit 1 anywhere the non-finite values are the data's own and the layers move as before
    finite_somewhere = any(w & 2 for w in words[1:])
    for slot in range(1, _RANGE_SLOTS):
        if words[slot] & 2 or (words[slot] & 4 and not finite_somewhere):
            for ref in _SLOT_OWNERS.get(slot, []):
                o = ref()
                if o is None:
                    continue
                if isinstance(o, PackedChain):
                    if not o.state.get("off"):
                        o.state["off"] = True
                        moved.append("chained pair %dx%d->%d->%d: two launches" % (o.K1, o.N1, o.N1, o.N2))
                    continue
                if isinstance(o, PackedBneck):
                    if not o.state.get("off"):
                        o.state["off"] = True
                        moved.append("fused bottleneck %d->64->256: separate launches" % o.cin)
                    continue
                tier = o.state["tier"]
                t_launch, one = o.state.get("at", {}).get(ws_key, (tier, o.last_one))
                new = 1 if (one and t_launch < 1) else 2
                if new > tier:
                    o.state["tier"] = new
                    moved.append("%dx%d %d->%d: |a| > %s -> %s" % (o.R, o.S, o.C, o.K, "4094" if one else "65504",
                                                                   "two accumulators" if new == 1 else "bf16x3"))
                elif tier > t_launch:
                    stale += 1      # raised by a pass launched before the layer was moved: that pass is run again, nothing moves
    if moved:
        RANGE_EPOCH += 1
        import logging

        logging.getLogger("lvc_amd").warning("activation range: %d layer(s) re-routed and kept there (%s); re-running the pass",
                                             len(moved), "; ".join(moved[:8]))
    e = Fp16RangeError("conv/GEMM kernel: an operand beyond the range of the fp16 split form it ran on (|a| > 4094 single-accumulator, "
                       "> 65504 two-accumulator, or NaN)" + ("; the layers concerned were moved to the next wider form" if moved else
                                                             "; set LVC_CONV_SPLIT=bf16x3 for range-free kernels"))
    e.rerouted = (bool(moved) or stale > 0) and not (words[0] & 6)
    raise e


class LaunchTimer:
    """Optional per-launch HIP-event bracket for the conv/GEMM kernel (bench.py's roofline leg).
    Events are recorded on the stream the kernel is launched on (torch's current stream)."""

    def __init__(self, only=None, every=1):
        self.records = []  # (algorithmic flops, start event, end event, engine)
        self.only = only   # None = bracket every launch; else the set of engine tags to bracket
        self.every = max(1, int(every))   # bracket the launches of every `every`-th step only (`next_step` counts them): an
        self.step = 0                     # event pair costs ~6 us of stream bubbles around a launch
        self.active = True

    def next_step(self):
        """Call at the start of a step: launches are bracketed in steps 0, every, 2*every, ..."""
        self.active = self.step % self.every == 0
        self.step += 1

    def steps_timed(self):
        return (self.step + self.every - 1) // self.every

    def algorithmic_bytes(self, engine=None):
        return sum(r[4] for r in self.records if (engine is None or r[3] == engine) and len(r) > 4)

    def flops_and_ms(self, engine=None):
        torch.cuda.synchronize()
        recs = [r for r in self.records if engine is None or r[3] == engine]
        fl = sum(r[0] for r in recs)
        ms = sum(r[1].elapsed_time(r[2]) for r in recs)
        return fl, ms, len(recs)


CONV_TIMER = None  # set to a LaunchTimer to instrument lvc_conv2d_nhwc_f32 launches
# Inner-product engine of the conv/GEMM layers with >= 128 output channels:
#   "bf16x3" = fp32-accurate 3-way bf16 operand split on the bf16 matrix cores (csrc/conv_bf16x3.hip)
#   "f32"    = v_mfma_f32_32x32x2_f32 (csrc/conv_igemm.hip); always used for the stem and the 64-channel layers
import os as _os

CONV_ENGINE = _os.environ.get("LVC_CONV_ENGINE", "bf16x3")
_BF16X3_MIN_K = 128
# 3x3 / stride 1 / pad 1 layers of the bf16x3 engine go to the halo kernel (csrc/conv3x3_halo.hip)
CONV_HALO = True       # False: the generic kernels (tests compare the two)
# BasicStem (conv 7x7/2 + FrozenBN + ReLU + max-pool 3x3/2) as one fused split-precision kernel (csrc/stem_pool.hip)
STEM_FUSED = True
_PW_NARROW = True
_PW_NARROW_MIN_C = 64
# operand split of the split-precision kernels that have both forms: "f16x2" = two fp16 planes, 3 MFMAs per block
# (Ootomo & Yokota; csrc/conv3x3_halo_h2.hip), "bf16x3" = three bf16 planes, 6 MFMAs per block (no range limit)
CONV_SPLIT = _os.environ.get("LVC_CONV_SPLIT", "f16x2")
# Operand split of the DATA-gradient convolutions.  "bf16x3" keeps fp32's exponent range (raw gradients of a mean-reduced
# loss sit around 1e-6 .. 1e-3, below fp16's normal range); "f16x2" is the faster two-way fp16 split and needs the
# gradients scaled into fp16's range first -- lvc_amd.solver.LossScaler does that with a power of two (exact) and switches
# this on for the backward pass it wraps.
DGRAD_SPLIT = _os.environ.get("LVC_DGRAD_SPLIT", "bf16x3")
# weight gradients without a loss scale: "bf16x3" = three-way bf16 split on the bf16 matrix cores, "f32" = fp32 MFMA
WGRAD_ENGINE = _os.environ.get("LVC_WGRAD_ENGINE", "bf16x3")
# inference: conv3 + stride-1 projection shortcut of res2.0 as one GEMM over [conv2 output | block input] (resnet.py)
FUSE_PROJECTION = _os.environ.get("LVC_FUSE_PROJECTION", "1") != "0"
_H2_PW_MIN_C = 64   # 64-channel streams too since the LDS-DMA kernel (0.32 -> 0.27 ms on res2 conv3)
# 3x3 fp16x2 layers of the FORWARD pass on the software-pipelined kernel (csrc/conv3x3_halo_s1.hip):
#   2 (default) = its single-accumulator form (row-scaled weight planes, activations x 2^4: |a| <= 4094; ~7 % faster on the 3x3
#       set -- the accumulate of the small cross products into the large sum costs the matrix pipe less power than a second full
#       accumulator -- at 1.7x the rounding noise of the two-accumulator form, 7.2e-8 of the output scale against 8e-8 .. 1.2e-7
#       for the reference's own fp32 CPU convolution, scripts/probe_halo_set.py), EXCEPT layers packed with `two_acc` (the RPN
#       head: its outputs feed top-k / NMS decisions and the post-trunk chain is held to the literal 1e-3, tests/test_gpu_chain.py);
#   1 = the numerics of conv3x3_halo_h2.hip everywhere (main + cross accumulators, same weight planes, |a| <= 65504);
#   0 = the round-1 kernel (conv3x3_halo_h2.hip), which data gradients (explicit `split`) always use.
HALO_S1 = 2
# pointwise fp16x2 layers with at least LVC_PW_S1_MIN_C input channels on the pipelined kernel (csrc/conv_pw_s1.hip): 2 = its
# single-accumulator form except `two_acc` layers, 1 = two accumulators everywhere, 0 = off (the LDS-DMA kernel for all of them)
PW_S1 = 2
_PW_S1_MIN_C = 64
_PW_S1_ONE_MIN_C = 256
_PW_S1_RES = 1     # 1: layers with a residual / upsample-add operand qualify too
# single-accumulator pointwise layers with >= 256 input channels and a multiple of 256 output channels on the 256 x 256 tile
# (csrc/conv_pw_w2.hip); 0 = the 256 x 128 tile of conv_pw_s1.hip for all of them
PW_W2 = _os.environ.get("LVC_PW_W2", "1") != "0"
# ... from 4096 input channels on (box-head fc1: 0.609 -> 0.599 / 0.593 -> 0.566 ms on two boxes).  Measured per layer in
# profiles/r05_pw_w2.txt: the stage loop is ~10 % faster per MFMA, but with 16 - 64 stages per tile (res4 / res5, the laterals) a tile's
# fill, hand-off and epilogue outweigh it -- and half as many, twice as large tiles split worse over 256 CUs -- so those stay on the 256 x 128 tile
_PW_W2_MIN_C = 4096
_PW_W2_MIN_ROWS = 4096
# inference: conv3 (+ shortcut add + ReLU) of a bottleneck and conv1 (+ ReLU) of the next one as ONE launch (csrc/conv_pw_chain.hip;
# modeling/backbone/resnet.py `BottleneckBlock.chain_to`); 0 = the two launches
CHAIN = _os.environ.get("LVC_CHAIN", "1") != "0"
_HALO_H2_MIN_TILES = 64    # smaller 3x3 layers (p6; p5 of fewer than seven images) use the bf16 kernels (tests set 0): scripts/probe_small_maps.py -- p5 of the batch of eight (80 tiles) 0.052 / 0.056 ms on the fp16-split kernel against 0.077 / 0.069, p6 (32 tiles) 0.041 / 0.050 against 0.040


# inference: a bottleneck's conv2 (3x3, direct single-accumulator kernel) hands its output to conv3 (pointwise, single accumulator) as the
# two fp16 planes conv3 multiplies -- conv3 skips the split (csrc/conv3x3_halo_s1.hip / conv_pw_s1.hip `_presplit`); bit-identical results.
# OFF by default: measured (scripts/probe_presplit.py, profiles/r06_presplit.txt) the pair of launches gets 2.2 % (res4) / 2.8 % (res5)
# shorter -- conv3 there moves 2.6 TB/s of fp32 rows, the split it skips was not what bounds it -- which is +0.15 % on the step
# (same-box A/B, four alternations, inside the noise): not worth a second hand-over format in the default path.  LVC_PRESPLIT=1 enables it.
PRESPLIT = _os.environ.get("LVC_PRESPLIT", "0") == "1"


def presplit_pair_ok(x, pc2, pc3, residual=None):
    """True where `conv2d_nhwc` would run pc2 on lvc_conv3x3_nhwc_f16s1 and pc3 on lvc_conv1x1_nhwc_f16s1 (both on tier 0): the pair
    `conv3x3_conv1x1_presplit` replaces launch for launch.  (The routing conditions of conv2d_nhwc, restated; the bit-identity test
    in tests/test_gpu_kernels.py fails if the two ever disagree.)"""
    if not (PRESPLIT and CONV_ENGINE == "bf16x3" and CONV_SPLIT == "f16x2" and CONV_HALO and HALO_S1 == 2 and PW_S1 == 2):
        return False
    N, H, W, C = x.shape
    rows = N * H * W
    return (pc2.mode == 0 and pc3.mode == 0 and C == pc2.C
            and pc2.R == 3 and pc2.S == 3 and pc2.stride == 1 and pc2.pad == 1 and pc2.C % 32 == 0 and pc2.K % 32 == 0 and pc2.K >= 64
            and pc3.R == 1 and pc3.S == 1 and pc3.stride == 1 and pc3.pad == 0 and pc3.C == pc2.K and pc3.K > 64 and pc3.K % 4 == 0
            and pc3.C >= max(_H2_PW_MIN_C, _PW_S1_MIN_C, _PW_S1_ONE_MIN_C)
            and pc2.state["tier"] == 0 and pc3.state["tier"] == 0 and not pc2.two_acc and not pc3.two_acc
            and N * ((H * W + 255) // 256) * ((pc2.K + 127) // 128) >= _HALO_H2_MIN_TILES
            and not (CONV_WINO and pc2.K >= 128 and wino_tiles(N, H, W, pc2.K) >= _WINO_MIN_TILES)
            and rows >= 2048 and rows * pc3.K < (1 << 29) and x.numel() < (1 << 29) and rows * pc2.K < (1 << 29)
            and (residual is None or (_PW_S1_RES and residual.shape[-1] % 4 == 0))
            and not (PW_W2 and pc3.C >= _PW_W2_MIN_C and pc3.K >= 256 and pc3.K % 256 == 0 and rows >= _PW_W2_MIN_ROWS))


def conv3x3_conv1x1_presplit(x, pc2, pc3, residual=None, relu=True):
    """relu(conv3x3(x) * s2 + t2) -> act(conv1x1(.) * s3 + t3 (+ residual)) where `presplit_pair_ok`: two launches, the tensor between
    them written as the second one's operand planes (include/lvc_amd.h, `_presplit`).  Reference resnet.py:200-212."""
    _req_cuda(x, residual)
    assert x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32
    N, H, W, C = x.shape
    mid = torch.empty(N, H, W, pc2.K, device=x.device, dtype=torch.float32)      # (planes: same bytes, not fp32 values)
    out = torch.empty(N, H, W, pc3.K, device=x.device, dtype=torch.float32)
    if residual is not None:
        assert residual.is_contiguous() and residual.dtype == torch.float32 and residual.shape[:3] == out.shape[:3]
    ldr = residual.shape[-1] if residual is not None else 0
    ws = ptr(conv_workspace(x.device))

    def timed(engine, flops, nbytes, launch):
        timer = CONV_TIMER
        if timer is not None and (not timer.active or (timer.only is not None and engine not in timer.only)):
            timer = None
        if timer is None:
            return launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        timer.records.append((flops, e0, e1, engine, nbytes))

    planes2, scale2 = pc2.split2s()
    planes3, scale3 = pc3.split2s()
    rows = N * H * W
    pc2.last_one = True
    pc3.last_one = True
    _lib.lib().lvc_set_range_slot(c_int(pc2.slot))
    timed("f16x2_halo", 2.0 * rows * pc2.K * C * 9, 4.0 * (rows * C + rows * pc2.K + pc2.K * C * 9),
          lambda: check(_lib.lib().lvc_conv3x3_nhwc_f16s1_presplit(
              ptr(x), ptr(planes2), ptr(scale2), ptr(pc2.shift), ptr(mid), c_int(N), c_int(H), c_int(W), c_int(C), c_int(pc2.K),
              c_int(pc2.Kg), c_int(pc3.slot), ws, _stream(x)), "lvc_conv3x3_nhwc_f16s1_presplit"))
    _lib.lib().lvc_set_range_slot(c_int(pc3.slot))
    timed("f16x2_pws1", 2.0 * rows * pc3.K * pc3.C, 4.0 * (rows * pc3.C + rows * pc3.K + (residual.numel() if residual is not None else 0) + pc3.K * pc3.C),
          lambda: check(_lib.lib().lvc_conv1x1_nhwc_f16s1_presplit(
              ptr(mid), ptr(planes3), ptr(scale3), ptr(pc3.shift), ptr(residual), ptr(out), c_int(N), c_int(H), c_int(W), c_int(pc3.C),
              c_int(pc3.K), c_int(1 if relu else 0), c_int(1 if residual is not None else 0), c_int(pc3.K), c_int(ldr), ws, _stream(x)),
              "lvc_conv1x1_nhwc_f16s1_presplit"))
    _lib.lib().lvc_set_range_slot(c_int(0))
    return out


def conv2d_nhwc(x, pc, relu=False, residual=None, res_mode=0, out=None, split=None, act=None):
    """x: [N,H,W,C] fp32 contiguous (NHWC).  Returns [N,Ho,Wo,K].
    act="gelu": torch.nn.GELU() (erf form) on the result -- in the epilogue of the LDS-DMA pointwise kernel where the layer
    runs on it (bit-identical to a separate lvc_gelu pass), as a second launch otherwise.
    split: None = the configured split (LVC_CONV_SPLIT); "bf16x3" keeps the fp32 exponent range (gradients).
    res_mode 1: residual has the output's shape; 2: residual is [N,Ho/2,Wo/2,K] and is
    nearest-x2-upsampled on the fly (FPN top-down path, reference fpn.py:131-133)."""
    _req_cuda(x, residual)
    assert x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32
    N, H, W, C = x.shape
    assert C == pc.C, "channel mismatch: tensor {} vs packed {}".format(C, pc.C)
    Ho = (H + 2 * pc.pad - pc.R) // pc.stride + 1
    Wo = (W + 2 * pc.pad - pc.S) // pc.stride + 1
    if out is None:
        out = torch.empty(N, Ho, Wo, pc.K, device=x.device, dtype=torch.float32)
    if residual is not None:
        assert residual.is_contiguous() and residual.dtype == torch.float32
        if res_mode == 0:
            res_mode = 1
    # The split-precision kernels address their operands through 32-bit buffer descriptors (< 2 GiB per tensor).  A batch whose
    # input / output / residual reaches 2^29 elements (32 images on the p2 map) is run in image groups that stay below it --
    # same kernels, same values (a tile never spans two images) -- instead of failing (logged once)
    big = max(x.numel(), out.numel(), residual.numel() if residual is not None else 0)
    if big >= (1 << 29) and N > 1 and out.is_contiguous() and out.shape[-1] == pc.K:
        per_image = (big + N - 1) // N
        nb = max(1, ((1 << 29) - 1) // per_image)
        _log_once("conv_batch_split", "conv/GEMM layer with %d elements in one tensor (batch %d): run in groups of %d images to stay "
                  "inside the kernels' 2 GiB buffer descriptors", big, N, nb)
        for n0 in range(0, N, nb):
            n1 = min(N, n0 + nb)
            conv2d_nhwc(x[n0:n1], pc, relu=relu, residual=residual[n0:n1] if residual is not None else None, res_mode=res_mode,
                        out=out[n0:n1], split=split, act=act)
        return out
    ldr = residual.shape[-1] if residual is not None else 0
    # this layer's range tier (check_conv_error_word): 1 = two accumulators, 2 = the range-free bf16x3 kernels
    tier = pc.state["tier"] if split is None else 0
    two_acc = pc.two_acc or tier >= 1
    if tier >= 2:
        split = "bf16x3"
    slotted = False
    engine = "f32"
    halo = CONV_HALO and pc.R == 3 and pc.S == 3 and pc.stride == 1 and pc.pad == 1 and pc.C % 32 == 0
    # narrow 1x1 layers (256 -> 64 reductions, the 15-channel RPN predictors) go to the 64- / 32-channel tiles of the
    # 256-row pointwise shape
    pw_narrow = _PW_NARROW and pc.R == 1 and pc.S == 1 and pc.pad == 0 and pc.C >= _PW_NARROW_MIN_C and pc.C % 32 == 0 and N * Ho * Wo >= 2048
    if (CONV_ENGINE == "bf16x3" and pc.mode == 0 and pc.K >= (64 if halo else 4 if pw_narrow else _BF16X3_MIN_K)
            and pc.K % 4 == 0 and out.shape[-1] % 4 == 0 and ldr % 4 == 0):
        h2_halo = halo and (split or CONV_SPLIT) == "f16x2" and N * ((H * W + 255) // 256) * ((pc.K + 127) // 128) >= _HALO_H2_MIN_TILES
        h2_pw = (not halo and (split or CONV_SPLIT) == "f16x2" and pc.R == 1 and pc.S == 1 and pc.pad == 0 and pc.C >= _H2_PW_MIN_C
                 and N * Ho * Wo >= 2048)   # the 256-row pointwise shape; 64-channel streams stay bf16x3 (f16x2 there: 0.312 vs 0.335 ms alone, no gain end to end)
        engine = "f16x2_halo" if h2_halo else "f16x2_pw" if h2_pw else "bf16x3_halo" if halo else "bf16x3"
        if engine.startswith("f16x2") and tier < 2 and split is None:
            _lib.lib().lvc_set_range_slot(c_int(pc.slot))     # this launch raises the LAYER's range word
            slotted = True
            pc.last_one = False
        if (engine == "f16x2_pw" and PW_S1 and split is None and pc.C >= _PW_S1_MIN_C and pc.K >= 64 and (residual is None or _PW_S1_RES)
                and out.numel() < (1 << 29)):
            engine = "f16x2_pws1"     # >= 64 input channels: the pipelined pointwise kernel (csrc/conv_pw_s1.hip)
    if (engine == "f16x2_halo" and HALO_S1 == 2 and split is None and tier == 0 and (not pc.two_acc or WINO_RPN) and CONV_WINO and residual is None and pc.K >= 128
            and out.is_contiguous() and out.shape[-1] == pc.K and wino_tiles(N, H, W, pc.K) >= _WINO_MIN_TILES):
        engine = "f16x2_wino"      # (decided before the timer's engine filter: bench.py brackets the launches of ONE engine in the timed region)
    timer = CONV_TIMER
    if timer is not None and (not timer.active or (timer.only is not None and engine not in timer.only)):
        timer = None
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    if engine != "f32":
        if engine == "f16x2_wino":
            # Winograd F(2,3) along x on the maps that fill the chip with its one-workgroup tiles (csrc/conv3x3_wino.hip)
            pc.last_one = True
            conv3x3_wino(x, pc, relu=relu, out=out)
        elif engine == "f16x2_halo" and HALO_S1 == 2 and split is None and not two_acc:
            planes, scale2 = pc.split2s()
            pc.last_one = True
            st = _lib.lib().lvc_conv3x3_nhwc_f16s1(
                ptr(x), ptr(planes), ptr(scale2), ptr(pc.shift), ptr(residual), ptr(out),
                c_int(N), c_int(H), c_int(W), c_int(C), c_int(pc.K), c_int(pc.Kg), c_int(1 if relu else 0),
                c_int(res_mode), c_int(out.shape[-1]), c_int(ldr), ptr(conv_workspace(x.device)), _stream(x))
            check(st, "lvc_conv3x3_nhwc_f16s1")
        elif engine == "f16x2_halo" and HALO_S1 >= 1 and split is None:
            st = _lib.lib().lvc_conv3x3_nhwc_f16x2_pipe(
                ptr(x), ptr(pc.split2h()), ptr(pc.scale), ptr(pc.shift), ptr(residual), ptr(out),
                c_int(N), c_int(H), c_int(W), c_int(C), c_int(pc.K), c_int(pc.Kg), c_int(1 if relu else 0),
                c_int(res_mode), c_int(out.shape[-1]), c_int(ldr), ptr(conv_workspace(x.device)), _stream(x))
            check(st, "lvc_conv3x3_nhwc_f16x2_pipe")
        elif engine == "f16x2_halo":
            st = _lib.lib().lvc_conv3x3_nhwc_f16x2(
                ptr(x), ptr(pc.split2h()), ptr(pc.scale), ptr(pc.shift), ptr(residual), ptr(out),
                c_int(N), c_int(H), c_int(W), c_int(C), c_int(pc.K), c_int(pc.Kg), c_int(1 if relu else 0),
                c_int(res_mode), c_int(out.shape[-1]), c_int(ldr), ptr(conv_workspace(x.device)), _stream(x))
            check(st, "lvc_conv3x3_nhwc_f16x2")
        elif halo:
            st = _lib.lib().lvc_conv3x3_nhwc_bf16x3(
                ptr(x), ptr(pc.split3()), ptr(pc.scale), ptr(pc.shift), ptr(residual), ptr(out),
                c_int(N), c_int(H), c_int(W), c_int(C), c_int(pc.K), c_int(pc.Kg), c_int(1 if relu else 0),
                c_int(res_mode), c_int(out.shape[-1]), c_int(ldr), ptr(conv_workspace(x.device)), _stream(x))
            check(st, "lvc_conv3x3_nhwc_bf16x3")
        elif engine == "f16x2_pws1":
            # pointwise layers with >= 64 input channels on the pipelined kernel (csrc/conv_pw_s1.hip: fc1 0.75 -> 0.59 ms, res4 / res5
            # conv1 -10..15 %, the memory-bound res2 / res3 conv3 -2..9 % since its epilogue stopped serialising rows:
            # scripts/probe_pw_set.py); precision policy as for the 3x3 layers, and the layers with < 256 input channels keep the
            # two-accumulator form -- bit-identical to the LDS-DMA kernel they ran on before (they are memory-bound: the form costs nothing)
            one = PW_S1 == 2 and not two_acc and C >= _PW_S1_ONE_MIN_C
            pc.last_one = one
            fused_act = act == "gelu" and not relu
            if fused_act:
                act = None
            code = c_int(2 if fused_act else 1 if relu else 0)
            if one and PW_W2 and C >= _PW_W2_MIN_C and pc.K >= 256 and pc.K % 256 == 0 and N * Ho * Wo >= _PW_W2_MIN_ROWS:
                # >= 256 input and output channels: the 256 x 256 tile (csrc/conv_pw_w2.hip; same operands, 0.67 x the bytes per MFMA)
                planes, scale2 = pc.split2s()
                st = _lib.lib().lvc_conv1x1_nhwc_f16s1_w2(
                    ptr(x), ptr(planes), ptr(scale2), ptr(pc.shift), ptr(residual), ptr(out), c_int(N), c_int(H), c_int(W), c_int(C),
                    c_int(pc.K), c_int(planes.shape[1]), c_int(pc.stride), code, c_int(res_mode), c_int(out.shape[-1]), c_int(ldr),
                    ptr(conv_workspace(x.device)), _stream(x))
            elif one:
                planes, scale2 = pc.split2s()
                st = _lib.lib().lvc_conv1x1_nhwc_f16s1(
                    ptr(x), ptr(planes), ptr(scale2), ptr(pc.shift), ptr(residual), ptr(out), c_int(N), c_int(H), c_int(W), c_int(C),
                    c_int(pc.K), c_int(pc.stride), code, c_int(res_mode), c_int(out.shape[-1]), c_int(ldr),
                    ptr(conv_workspace(x.device)), _stream(x))
            else:
                st = _lib.lib().lvc_conv1x1_nhwc_f16x2_pipe(
                    ptr(x), ptr(pc.split2h()), ptr(pc.scale), ptr(pc.shift), ptr(residual), ptr(out), c_int(N), c_int(H), c_int(W), c_int(C),
                    c_int(pc.K), c_int(pc.stride), code, c_int(res_mode), c_int(out.shape[-1]), c_int(ldr),
                    ptr(conv_workspace(x.device)), _stream(x))
            check(st, "lvc_conv1x1_nhwc_f16s1" if one else "lvc_conv1x1_nhwc_f16x2_pipe")
        elif engine == "f16x2_pw":
            # the LDS-DMA kernel addresses outputs / residuals through 32-bit buffer descriptors (< 2^29 elements) and moves
            # residual rows in 32-channel chunks; anything else stays on the register-staged kernel (logged once)
            dma_ok = out.numel() < (1 << 29) and (residual is None or (residual.numel() < (1 << 29) and pc.K % 32 == 0))
            if not dma_ok:
                _log_once("pw_dma_fallback", "pointwise layer %dx%d->%d (%d output elements, residual %s) is outside the LDS-DMA "
                          "kernel's range; it runs on the register-staged fp16x2 kernel", N * H * W, C, pc.K, out.numel(),
                          "yes" if residual is not None else "no")
            fn = "lvc_conv2d_nhwc_f16x2_dma" if dma_ok else "lvc_conv2d_nhwc_f16x2"
            fused_act = act == "gelu" and fn.endswith("_dma") and not relu
            if fused_act:
                act = None
            st = getattr(_lib.lib(), fn)(
                ptr(x), ptr(pc.split2h()), ptr(pc.scale), ptr(pc.shift), ptr(residual), ptr(out),
                c_int(N), c_int(H), c_int(W), c_int(C), c_int(pc.K), c_int(pc.R), c_int(pc.S),
                c_int(pc.stride), c_int(pc.pad), c_int(pc.Kg), c_int(2 if fused_act else 1 if relu else 0), c_int(res_mode),
                c_int(out.shape[-1]), c_int(ldr), ptr(conv_workspace(x.device)), _stream(x))
            check(st, "lvc_conv2d_nhwc_f16x2")
        else:
            st = _lib.lib().lvc_conv2d_nhwc_bf16x3(
                ptr(x), ptr(pc.split3()), ptr(pc.scale), ptr(pc.shift), ptr(residual), ptr(out),
                c_int(N), c_int(H), c_int(W), c_int(C), c_int(pc.K), c_int(pc.R), c_int(pc.S),
                c_int(pc.stride), c_int(pc.pad), c_int(pc.Kg), c_int(1 if relu else 0), c_int(res_mode),
                c_int(out.shape[-1]), c_int(ldr), ptr(conv_workspace(x.device)), _stream(x))
            check(st, "lvc_conv2d_nhwc_bf16x3")
    else:
        st = _lib.lib().lvc_conv2d_nhwc_f32(
            ptr(x), ptr(pc.w), ptr(pc.scale), ptr(pc.shift), ptr(residual), ptr(out),
            c_int(N), c_int(H), c_int(W), c_int(C), c_int(pc.K), c_int(pc.R), c_int(pc.S),
            c_int(pc.stride), c_int(pc.pad), c_int(pc.Kg), c_int(1 if relu else 0), c_int(res_mode),
            c_int(out.shape[-1]), c_int(ldr), c_int(pc.mode), ptr(conv_workspace(x.device)), _stream(x))
        check(st, "lvc_conv2d_nhwc_f32")
    if slotted:
        _lib.lib().lvc_set_range_slot(c_int(0))
    if timer is not None:
        e1.record()
        c_real = 3 if pc.mode == 1 else C
        # algorithmic bytes of the launch: the input pixels it needs, its output, the residual operand, the weights -- each once
        nbytes = 4.0 * (N * (Ho * Wo if (pc.R == 1 and pc.S == 1) else H * W) * c_real + N * Ho * Wo * pc.K
                        + (residual.numel() if residual is not None else 0) + pc.K * c_real * pc.R * pc.S)
        timer.records.append((2.0 * N * Ho * Wo * pc.K * c_real * pc.R * pc.S, e0, e1, engine, nbytes))
    if act == "gelu":
        check(_lib.lib().lvc_gelu(ptr(out), ptr(out), c_longlong(out.numel()), _stream(out)), "lvc_gelu")
    elif act is not None:
        raise ValueError("unknown activation {!r}".format(act))
    return out


class PackedChain:
    """Weights of two chained pointwise layers for lvc_conv1x1_chain_nhwc_f16s1 (csrc/conv_pw_chain.hip)."""

    __slots__ = ("wa", "sa", "ta", "wb", "sb", "tb", "K1", "N1", "N2", "slot", "state", "__weakref__")


_PERM16 = (0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15)
CHAIN_SHAPES = {(64, 256, 64), (128, 256, 64), (128, 512, 128)}


def _chain_planes(pc):
    """Row-scaled fp16 planes of a pointwise layer's packed weights with the contraction index permuted within every 16
    (position p holds channel 16 * (p // 16) + _PERM16[p % 16]): the order in which an accumulator lane of the transposed MFMA
    holds a block's channels (csrc/conv_pw_chain.hip)."""
    assert pc.R == 1 and pc.S == 1 and pc.mode == 0 and pc.stride == 1 and pc.pad == 0
    rows, Kg = pc.w.shape
    idx = torch.arange(Kg, device=pc.w.device).view(-1, 16)[:, list(_PERM16)].reshape(-1)
    wperm = pc.w[:, idx].contiguous()
    planes = torch.empty((2, rows, Kg), device=pc.w.device, dtype=torch.float16)
    fac = torch.empty(rows, device=pc.w.device, dtype=torch.float32)
    check(_lib.lib().lvc_split_weights_rowscaled(ptr(wperm), c_int(rows), c_int(Kg), ptr(planes), ptr(fac), _stream(pc.w)),
          "lvc_split_weights_rowscaled")
    fac = fac[: pc.K]
    return planes, (fac * pc.scale if pc.scale is not None else fac).contiguous()


# 3x3 / stride 1 single-accumulator layers as Winograd F(2,3) along x (csrc/conv3x3_wino.hip: two thirds of the MFMAs at the direct
# evaluation's fp32 error, scripts/winograd_error.py) where the launch has at least _WINO_MIN_TILES 256-pixel x 128-channel tiles --
# the kernel is one workgroup per tile, no stream-K: small maps keep the direct kernel
CONV_WINO = _os.environ.get("LVC_CONV_WINO", "1") != "0"
# the Winograd kernel's work distribution: 0 = one workgroup per tile (default: fastest on every routed layer, csrc/conv3x3_wino.hip),
# 1 = stream-K, 2 = persistent workgroups on whole tiles
WINO_STREAMK = int(_os.environ.get("LVC_WINO_STREAMK", "0"))


def set_wino_streamk(mode):
    global WINO_STREAMK
    WINO_STREAMK = int(mode)
    _lib.lib().lvc_set_wino_streamk(c_int(int(mode)))
WINO_RPN = True        # the RPN head's large levels too (kernels.conv3x3_levels_pred); False: the two-accumulator direct kernel for all levels
_WINO_MIN_TILES = 512


def pack_wino(pc):
    """Transformed, row-scaled fp16 weight planes of a packed 3x3 layer for lvc_conv3x3_nhwc_wino, cached on the PackedConv:
    U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2 per filter row (formed in fp64, one fp32 rounding), split by
    lvc_split_weights_rowscaled (rows scaled by a power of two so that neither plane leaves fp16's normal range), stored
    [3 rows][C/16][4 positions][2 planes][Kpad][16].  -> (planes uint16, scale [K] = row factor x the layer's scale)."""
    cached = pc.state.get("_wino")
    if cached is not None and cached[0] is pc.w:
        return cached[1], cached[2]
    assert pc.R == 3 and pc.S == 3 and pc.mode == 0 and pc.C % 32 == 0
    K, C = pc.K, pc.C
    w = pc.w[:K].view(K, C // 32, 3, 3, 32).permute(0, 1, 4, 2, 3).reshape(K, C, 3, 3).double()       # OIHW from the (c/32, r, s, c%32) packing
    g0, g1, g2 = w[..., 0], w[..., 1], w[..., 2]
    U = torch.stack([g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2], dim=-1).float()                 # [K, C, 3 rows, 4 positions]
    Kpad = (K + 127) // 128 * 128
    flat = torch.zeros(Kpad, C * 12, device=pc.w.device)
    flat[:K] = U.reshape(K, C * 12)
    planes = torch.empty(2, Kpad, C * 12, dtype=torch.int16, device=pc.w.device)
    fac = torch.empty(Kpad, device=pc.w.device)
    check(_lib.lib().lvc_split_weights_rowscaled(ptr(flat), c_int(Kpad), c_int(C * 12), ptr(planes), ptr(fac), _stream(flat)), "lvc_split_weights_rowscaled")
    u = planes.view(2, Kpad, C // 16, 16, 3, 4).permute(4, 2, 5, 0, 1, 3).contiguous()                 # [r][kc][p][plane][Kpad][16]
    scale = (fac[:K] * pc.scale if pc.scale is not None else fac[:K]).contiguous()
    pc.state["_wino"] = (pc.w, u, scale)
    return u, scale


def conv3x3_wino(x, pc, relu=False, out=None):
    """y = act(conv3x3(x) * scale + shift) on the Winograd F(2,3) kernel.  x [N,H,W,C] fp32 NHWC contiguous."""
    _req_cuda(x)
    N, H, W, C = x.shape
    u, scale = pack_wino(pc)
    if out is None:
        out = torch.empty(N, H, W, pc.K, device=x.device, dtype=torch.float32)
    check(_lib.lib().lvc_conv3x3_nhwc_wino(ptr(x), ptr(u), ptr(scale), ptr(pc.shift), ptr(out), c_int(N), c_int(H), c_int(W), c_int(C), c_int(pc.K),
                                           c_int(u.shape[4]), c_int(1 if relu else 0), c_int(out.stride(2)), ptr(conv_workspace(x.device)), _stream(x)),
          "lvc_conv3x3_nhwc_wino")
    return out


def wino_tiles(N, H, W, K):
    return N * ((H + 7) // 8) * ((W + 31) // 32) * ((K + 127) // 128)


_GROUP_SLOTS = {}


def _group_slot(pcs):
    """One range slot owned by all layers of a grouped launch: a raised word moves every one of them to the next wider form."""
    key = tuple(id(pc.state) for pc in pcs)
    ent = _GROUP_SLOTS.get(key)
    if ent is None or any(r() is None for r in ent[1]):
        import weakref

        slot = _new_range_slot(pcs[0])
        for pc in pcs[1:]:
            _SLOT_OWNERS[slot].append(weakref.ref(pc))
        if len(_GROUP_SLOTS) > 64:
            _GROUP_SLOTS.clear()
        ent = _GROUP_SLOTS[key] = (slot, [weakref.ref(pc) for pc in pcs])
    else:
        # the packed layers may have been rebuilt (new objects, same range state): keep the slot's owners current
        import weakref

        # (merged with the live owners: a slot handed out twice after the round-robin wrapped keeps its other layer)
        live = [r for r in _SLOT_OWNERS.get(ent[0], []) if r() is not None and all(r() is not pc for pc in pcs)]
        _SLOT_OWNERS[ent[0]] = live + [weakref.ref(pc) for pc in pcs]
    return ent[0]


def conv3x3_levels(xs, pc, relu=False, outs=None):
    """A 3x3 / stride 1 / pad 1 layer over several maps xs[l] [N,H_l,W_l,C] in ONE launch of the pipelined fp16-split kernel (one
    stream-K split over the row tiles of all maps: the small maps no longer pay a launch each that cannot fill the chip).  `pc`: one
    PackedConv -- the SAME layer on every map (the RPN head over the pyramid levels, lvc_conv3x3_nhwc_f16_levels) -- or a list with a
    layer per map, all of one shape and precision form (the FPN output convs, lvc_conv3x3_nhwc_f16_layers).  outs[l] (optional):
    [N,H_l,W_l,K] contiguous buffers to write.  Falls back to one conv2d_nhwc call per map where that kernel is not the layers'
    route (another engine or split, a range tier on the bf16x3 kernels, mixed forms, maps of 2 GiB)."""
    _req_cuda(*xs)
    pcs = list(pc) if isinstance(pc, (list, tuple)) else [pc] * len(xs)
    shared = not isinstance(pc, (list, tuple))
    p0 = pcs[0]
    N, C = xs[0].shape[0], xs[0].shape[3]
    if outs is None:
        outs = [torch.empty(x.shape[0], x.shape[1], x.shape[2], q.K, device=x.device, dtype=torch.float32) for x, q in zip(xs, pcs)]
    forms = {(q.two_acc or q.state["tier"] >= 1) for q in pcs}
    ok = (len(xs) > 1 and len(xs) <= 6 and len(pcs) == len(xs) and CONV_ENGINE == "bf16x3" and CONV_SPLIT == "f16x2" and CONV_HALO and HALO_S1 >= 1
          and len(forms) == 1 and all(q.state["tier"] < 2 and q.mode == 0 and q.R == 3 and q.S == 3 and q.stride == 1 and q.pad == 1 and q.C == C
                                      and q.K == p0.K and q.Kg == p0.Kg for q in pcs)
          and C % 32 == 0 and p0.K >= 64 and p0.K % 4 == 0
          and all(x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32 and x.shape[0] == N and x.shape[3] == C
                  and x.numel() < (1 << 29) for x in xs)
          and all(o.is_contiguous() and o.shape == (x.shape[0], x.shape[1], x.shape[2], p0.K) for o, x in zip(outs, xs))
          and N * ((max(x.shape[1] * x.shape[2] for x in xs) + 255) // 256) * ((p0.K + 127) // 128) >= _HALO_H2_MIN_TILES)
    if not ok:
        for x, q, o in zip(xs, pcs, outs):
            conv2d_nhwc(x, q, relu=relu, out=o)
        return outs
    one = HALO_S1 == 2 and not (True in forms)
    # (`two_acc` layers -- the RPN head outside its fused-predictor launch, e.g. in a training forward -- qualify under WINO_RPN: the
    # Winograd form's error is below the one-accumulator direct form's and passes the head's chain test, see conv3x3_levels_pred)
    if CONV_WINO and HALO_S1 == 2 and all(q.state["tier"] == 0 and (not q.two_acc or WINO_RPN) for q in pcs):
        # maps large enough to fill the chip with one-workgroup tiles run on the Winograd F(2,3) kernel (two thirds of the MFMAs), each
        # alone; the small ones stay one grouped launch of the direct kernel
        big = [i for i, (x, q) in enumerate(zip(xs, pcs)) if wino_tiles(x.shape[0], x.shape[1], x.shape[2], q.K) >= _WINO_MIN_TILES and q.K >= 128]
        if big:
            for i in big:
                conv2d_nhwc(xs[i], pcs[i], relu=relu, out=outs[i])
            rest = [i for i in range(len(xs)) if i not in big]
            if rest:
                conv3x3_levels([xs[i] for i in rest], pc if shared else [pcs[i] for i in rest], relu=relu, outs=[outs[i] for i in rest])
            return outs
    L = len(xs)
    timer = CONV_TIMER
    if timer is not None and (not timer.active or (timer.only is not None and "f16x2_halo" not in timer.only)):
        timer = None
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    XP, IP = c_void_p * L, c_int * L
    xp, yp = XP(*[x.data_ptr() for x in xs]), XP(*[o.data_ptr() for o in outs])
    hs, ws = IP(*[x.shape[1] for x in xs]), IP(*[x.shape[2] for x in xs])
    for q in pcs:
        q.last_one = one
    ops = [q.split2s() if one else (q.split2h(), q.scale) for q in (pcs[:1] if shared else pcs)]
    _lib.lib().lvc_set_range_slot(c_int(p0.slot if shared else _group_slot(pcs)))
    if shared:
        st = _lib.lib().lvc_conv3x3_nhwc_f16_levels(c_int(1 if one else 0), xp, yp, hs, ws, c_int(L), ptr(ops[0][0]), ptr(ops[0][1]), ptr(p0.shift),
                                                    c_int(N), c_int(C), c_int(p0.K), c_int(p0.Kg), c_int(1 if relu else 0),
                                                    ptr(conv_workspace(xs[0].device)), _stream(xs[0]))
    else:
        def table(ts):
            return XP(*[0 if t is None else t.data_ptr() for t in ts])
        st = _lib.lib().lvc_conv3x3_nhwc_f16_layers(c_int(1 if one else 0), xp, yp, hs, ws, c_int(L), table([o[0] for o in ops]),
                                                    table([o[1] for o in ops]), table([q.shift for q in pcs]),
                                                    c_int(N), c_int(C), c_int(p0.K), c_int(p0.Kg), c_int(1 if relu else 0),
                                                    ptr(conv_workspace(xs[0].device)), _stream(xs[0]))
    _lib.lib().lvc_set_range_slot(c_int(0))
    check(st, "lvc_conv3x3_nhwc_f16_levels")
    if timer is not None:
        e1.record()
        px = sum(x.shape[0] * x.shape[1] * x.shape[2] for x in xs)
        timer.records.append((2.0 * px * p0.K * C * 9, e0, e1, "f16x2_halo", 4.0 * (px * C + px * p0.K + (1 if shared else L) * p0.K * C * 9)))
    return outs


def conv3x3_levels_pred(xs, pc, pred, relu=True, _outs=None):
    """act(conv3x3(xs[l])) through the pointwise layer `pred` (<= 32 outputs) in the ONE launch of `conv3x3_levels`: the hidden maps are
    never written -- every workgroup of the 3x3 kernel contracts its 128 hidden channels with the pointwise weights in its epilogue and
    adds the slice to the (zeroed) output atomically (two slices per element: order-free, csrc/conv3x3_halo_s1.hip).  The RPN head:
    conv + ReLU, then objectness | anchor deltas.  Returns the list of [N,H_l,W_l,pred.K] outputs (views of one [sum of pixels, pred.K]
    buffer), or None where the launch does not apply (the caller runs the two layers one after the other)."""
    _req_cuda(*xs)
    N, C = xs[0].shape[0], xs[0].shape[3]
    ok = (1 <= len(xs) <= 6 and CONV_ENGINE == "bf16x3" and CONV_SPLIT == "f16x2" and CONV_HALO and HALO_S1 >= 1
          and pc.state["tier"] < 2 and (pc.two_acc or pc.state["tier"] == 1 or HALO_S1 == 1)      # the two-accumulator instance
          and pc.mode == 0 and pc.R == 3 and pc.S == 3 and pc.stride == 1 and pc.pad == 1 and pc.C == C and C % 32 == 0
          and pc.K in (128, 256) and pc.Kg == 9 * C
          and pred.mode == 0 and pred.R == 1 and pred.S == 1 and pred.stride == 1 and pred.pad == 0 and pred.C == pc.K and pred.Kg == pc.K
          and 1 <= pred.K <= 32 and pred.w.shape[0] >= 32 and pred.state["tier"] < 2
          and all(x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32 and x.shape[0] == N and x.shape[3] == C
                  and x.numel() < (1 << 29) for x in xs)
          and N * ((max(x.shape[1] * x.shape[2] for x in xs) + 255) // 256) * ((pc.K + 127) // 128) >= _HALO_H2_MIN_TILES)
    if not ok:
        return None
    ms = [x.shape[0] * x.shape[1] * x.shape[2] for x in xs]
    if _outs is None:
        y = torch.zeros(sum(ms), pred.K, device=xs[0].device, dtype=torch.float32)
        outs, off = [], 0
        for x, m in zip(xs, ms):
            outs.append(y[off:off + m].view(x.shape[0], x.shape[1], x.shape[2], pred.K))
            off += m
    else:
        outs = _outs
    if CONV_WINO and WINO_RPN and HALO_S1 == 2 and pc.state["tier"] == 0:
        # The levels that fill the chip with one-workgroup tiles on the Winograd kernel, each alone, with the same epilogue; the small
        # ones stay one grouped launch of the two-accumulator direct kernel.  (The head is packed `two_acc`: its logits decide top-k and
        # NMS.  The Winograd form's measured error lies between the two direct forms' -- rms 4.8e-8 of the output scale against 6.7e-8
        # one accumulator / ~4e-8 two -- and the post-trunk chain test holds it to the same logit bar: tests/test_gpu_chain.py.)
        big = [i for i, x in enumerate(xs) if wino_tiles(x.shape[0], x.shape[1], x.shape[2], pc.K) >= _WINO_MIN_TILES]
        if big:
            u, scale = pack_wino(pc)
            pplanes = pred.split2h()
            pc.last_one = True
            pred.last_one = False
            wt = CONV_TIMER
            if wt is not None and (not wt.active or (wt.only is not None and "f16x2_wino" not in wt.only)):
                wt = None
            for i in big:
                x = xs[i]
                if wt is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                _lib.lib().lvc_set_range_slot(c_int(pc.slot))
                st = _lib.lib().lvc_conv3x3_nhwc_wino_pred(ptr(x), ptr(u), ptr(scale), ptr(pc.shift), ptr(outs[i]), c_int(N), c_int(x.shape[1]), c_int(x.shape[2]),
                                                          c_int(C), c_int(pc.K), c_int(u.shape[4]), c_int(1 if relu else 0), c_int(pred.K), ptr(pplanes),
                                                          ptr(pred.scale), ptr(pred.shift), c_int(pred.K), c_int(pred.w.shape[0]), c_int(pred.slot),
                                                          ptr(conv_workspace(x.device)), _stream(x))
                _lib.lib().lvc_set_range_slot(c_int(0))
                check(st, "lvc_conv3x3_nhwc_wino_pred")
                if wt is not None:
                    e1.record()
                    px = x.shape[0] * x.shape[1] * x.shape[2]
                    wt.records.append((2.0 * px * pc.K * (C * 9 + pred.K), e0, e1, "f16x2_wino", 4.0 * (px * C + px * pred.K + pc.K * (C * 9 + pred.K))))
            rest = [i for i in range(len(xs)) if i not in big]
            if rest:
                sub = conv3x3_levels_pred([xs[i] for i in rest], pc, pred, relu=relu, _outs=[outs[i] for i in rest])
                assert sub is not None
            return outs
    L = len(xs)
    timer = CONV_TIMER
    if timer is not None and (not timer.active or (timer.only is not None and "f16x2_halo" not in timer.only)):
        timer = None
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    XP, IP = c_void_p * L, c_int * L
    xp, yp = XP(*[x.data_ptr() for x in xs]), XP(*[o.data_ptr() for o in outs])
    hs, ws = IP(*[x.shape[1] for x in xs]), IP(*[x.shape[2] for x in xs])
    pc.last_one = False
    pred.last_one = False
    planes, pplanes = pc.split2h(), pred.split2h()
    _lib.lib().lvc_set_range_slot(c_int(pc.slot))
    st = _lib.lib().lvc_conv3x3_nhwc_f16_levels_pred(c_int(0), xp, yp, hs, ws, c_int(L), ptr(planes), ptr(pc.scale), ptr(pc.shift), c_int(N),
                                                     c_int(C), c_int(pc.K), c_int(pc.Kg), c_int(1 if relu else 0), ptr(pplanes),
                                                     ptr(pred.scale), ptr(pred.shift), c_int(pred.K), c_int(pred.w.shape[0]),
                                                     c_int(pred.K), c_int(pred.slot), ptr(conv_workspace(xs[0].device)), _stream(xs[0]))
    _lib.lib().lvc_set_range_slot(c_int(0))
    check(st, "lvc_conv3x3_nhwc_f16_levels_pred")
    if timer is not None:
        e1.record()
        px = sum(ms)
        timer.records.append((2.0 * px * pc.K * (C * 9 + pred.K), e0, e1, "f16x2_halo", 4.0 * (px * C + px * pred.K + pc.K * (C * 9 + pred.K))))
    return outs


def pack_chain(pc_a, pc_b, state=None):
    """pc_a: the first pointwise layer (K1 -> N1), pc_b: the second (N1 -> N2), both `pack_conv` results of 1x1 layers.
    state: a dict the owner keeps across re-packs; state["off"] is set when the pair overflowed the kernel's range
    (|a| > 4094 for its input or the intermediate: `check_conv_error_word`) and must run as two launches from then on."""
    assert pc_b.C == pc_a.K
    ch = PackedChain()
    ch.slot = _new_range_slot(ch)
    ch.state = state if state is not None else {"off": False}
    ch.wa, ch.sa = _chain_planes(pc_a)
    ch.wb, ch.sb = _chain_planes(pc_b)
    ch.ta, ch.tb = pc_a.shift, pc_b.shift
    ch.K1, ch.N1, ch.N2 = pc_a.C, pc_a.K, pc_b.K
    return ch


def conv1x1_chain(x, ch, residual=None, relu1=True, relu2=True, out1=None, out2=None):
    """y1 = act1(x Wa^T * sa + ta (+ residual)), y2 = act2(y1 Wb^T * sb + tb) in one launch; x [..., ldx] NHWC whose first K1
    channels are the first layer's input (a row stride wider than K1 is allowed).  Returns (y1, y2)."""
    _req_cuda(x, residual)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.shape[-1] >= ch.K1
    M = x.numel() // x.shape[-1]
    lead = tuple(x.shape[:-1])
    if out1 is None:
        out1 = torch.empty(lead + (ch.N1,), device=x.device, dtype=torch.float32)
    if out2 is None:
        out2 = torch.empty(lead + (ch.N2,), device=x.device, dtype=torch.float32)
    if residual is not None:
        assert residual.is_contiguous() and residual.dtype == torch.float32 and residual.numel() // residual.shape[-1] == M
    timer = CONV_TIMER
    if timer is not None and (not timer.active or (timer.only is not None and "f16s1_chain" not in timer.only)):
        timer = None
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.lib().lvc_set_range_slot(c_int(ch.slot))
    st = _lib.lib().lvc_conv1x1_chain_nhwc_f16s1(
        ptr(x), c_int(x.shape[-1]), ptr(ch.wa), c_int(ch.wa.shape[1]), ptr(ch.sa), ptr(ch.ta), ptr(residual),
        c_int(residual.shape[-1] if residual is not None else 0), ptr(out1), c_int(out1.shape[-1]), c_int(1 if relu1 else 0),
        ptr(ch.wb), c_int(ch.wb.shape[1]), ptr(ch.sb), ptr(ch.tb), ptr(out2), c_int(out2.shape[-1]), c_int(1 if relu2 else 0),
        c_int(M), c_int(ch.K1), c_int(ch.N1), c_int(ch.N2), ptr(conv_workspace(x.device)), _stream(x))
    _lib.lib().lvc_set_range_slot(c_int(0))
    check(st, "lvc_conv1x1_chain_nhwc_f16s1")
    if timer is not None:
        e1.record()
        nbytes = 4.0 * (M * (ch.K1 + ch.N1 + ch.N2 + (ch.N1 if residual is not None else 0)) + ch.K1 * ch.N1 + ch.N1 * ch.N2)
        timer.records.append((2.0 * M * (ch.K1 * ch.N1 + ch.N1 * ch.N2), e0, e1, "f16s1_chain", nbytes))
    return out1, out2


class PackedBneck:
    """One bottleneck block's weights as the stage images of lvc_bottleneck_nhwc_f16s1 (csrc/conv_bneck.hip)."""

    __slots__ = ("w", "s1", "t1", "s2", "t2", "s3", "t3", "cin", "proj", "slot", "state", "__weakref__")


BNECK = _os.environ.get("LVC_BNECK", "1") != "0"      # whole res2 blocks (64 mid, 256 out channels, stride 1) as one launch
BNECK_SHAPES = {(256, 64, 256, False), (64, 64, 256, True)}      # (in, mid, out channels, projection shortcut)
_BNECK_IDX = {}


def _bneck_index(cin, proj, device):
    """Gather indices (plane, row, column) into the three layers' [2][rows][Kg] split planes for every fp16 of the stage images:
    a stage is a run of 1 KB fragments in lane order -- lane l holds row l % 32 of the fragment's 32-row block and the 8
    contraction entries of k half l // 32 in the permuted order of csrc/conv_pw_chain.hip (`_PERM16`)."""
    key = (cin, proj, str(device))
    if key in _BNECK_IDX:
        return _BNECK_IDX[key]
    import numpy as np

    lane = np.arange(64)
    rowl = (lane % 32)[:, None].repeat(8, 1)                                    # [64, 8]
    kpos = np.asarray(_PERM16)[(lane // 32)[:, None] * 8 + np.arange(8)[None]]   # natural k16 offset held at (lane, slot)

    def frags(specs):
        pl = np.stack([np.full((64, 8), s[0]) for s in specs])
        rows = np.stack([s[1] + rowl for s in specs])
        cols = np.stack([np.asarray(s[2])[kpos] for s in specs])
        return pl, rows, cols

    a16 = np.arange(16)
    p1 = [(pl, 32 * cb, 32 * c + 16 * sp + a16) for c in range(cin // 32) for sp in range(2) for cb in range(2) for pl in range(2)]

    def col2(ch, r, s_):      # pack_conv's contraction order: (c // 32, r, s, c % 32)
        return (ch // 32) * 288 + (r * 3 + s_) * 32 + ch % 32

    # conv2: the 36 (k16 step, tap) pairs in the order the kernel walks them, two a stage
    p2 = [(pl, 32 * cb, col2(16 * (u // 9) + a16, (u % 9) // 3, (u % 9) % 3)) for u in range(36) for cb in range(2) for pl in range(2)]
    p3 = [(pl, 32 * j, 64 * half + 16 * s + a16) for j in range(8) for half in range(2 if proj else 1) for s in range(4)
          for pl in range(2)]
    out = tuple(tuple(torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in frags(sp)) for sp in (p1, p2, p3))
    _BNECK_IDX[key] = out
    return out


def pack_bottleneck(pc1, pc2, pc3, proj, state=None):
    """pc1 / pc2 / pc3: `pack_conv` results of a bottleneck's conv1 (1x1), conv2 (3x3, pad 1) and conv3 (1x1; with proj the fused
    [conv3 | projection shortcut] pack of BottleneckBlock._fused_projection, 128 contraction channels), FrozenBN folded."""
    assert pc1.R == 1 and pc2.R == 3 and pc2.S == 3 and pc2.pad == 1 and pc2.stride == 1 and pc3.R == 1 and pc1.stride == 1
    cin = pc1.C
    assert (cin, pc1.K, pc3.K, bool(proj)) in BNECK_SHAPES and pc2.C == 64 and pc2.K == 64 and pc3.C == (128 if proj else 64)
    dev = pc1.w.device
    bk = PackedBneck()
    bk.slot = _new_range_slot(bk)
    bk.state = state if state is not None else {"off": False}
    bk.cin, bk.proj = cin, bool(proj)
    parts = []
    affine = []
    for pc, (pl, rows, cols) in zip((pc1, pc2, pc3), _bneck_index(cin, bool(proj), dev)):
        nrows, Kg = pc.w.shape
        planes = torch.empty((2, nrows, Kg), device=dev, dtype=torch.float16)
        fac = torch.empty(nrows, device=dev, dtype=torch.float32)
        check(_lib.lib().lvc_split_weights_rowscaled(ptr(pc.w), c_int(nrows), c_int(Kg), ptr(planes), ptr(fac), _stream(pc.w)),
              "lvc_split_weights_rowscaled")
        parts.append(planes[pl, rows, cols].reshape(-1))
        fac = fac[: pc.K]
        affine.append(((fac * pc.scale) if pc.scale is not None else fac).contiguous())
        affine.append(pc.shift.contiguous() if pc.shift is not None else torch.zeros(pc.K, device=dev, dtype=torch.float32))
    bk.w = torch.cat(parts).contiguous()
    bk.s1, bk.t1, bk.s2, bk.t2, bk.s3, bk.t3 = affine
    return bk


def bottleneck_fused(x, bk, out=None):
    """relu(bn3(conv3(relu(bn2(conv2(relu(bn1(conv1(x)))))))) + shortcut(x)) in ONE launch (reference resnet.py:195-211);
    x [N,H,W,ldx] NHWC fp32 whose first bk.cin channels are the block's input."""
    _req_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and x.shape[-1] >= bk.cin
    N, H, W, ldx = x.shape
    if out is None:
        out = torch.empty(N, H, W, 256, device=x.device, dtype=torch.float32)
    timer = CONV_TIMER
    if timer is not None and (not timer.active or (timer.only is not None and "f16s1_bneck" not in timer.only)):
        timer = None
    if timer is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    _lib.lib().lvc_set_range_slot(c_int(bk.slot))
    st = _lib.lib().lvc_bottleneck_nhwc_f16s1(ptr(x), c_int(ldx), ptr(out), c_int(out.shape[-1]), c_int(N), c_int(H), c_int(W),
                                              c_int(bk.cin), c_int(1 if bk.proj else 0), ptr(bk.w), ptr(bk.s1), ptr(bk.t1), ptr(bk.s2),
                                              ptr(bk.t2), ptr(bk.s3), ptr(bk.t3), ptr(conv_workspace(x.device)), _stream(x))
    _lib.lib().lvc_set_range_slot(c_int(0))
    check(st, "lvc_bottleneck_nhwc_f16s1")
    if timer is not None:
        e1.record()
        M = N * H * W
        kk = bk.cin * 64 + 64 * 64 * 9 + 64 * 256 + (bk.cin * 256 if bk.proj else 0)
        timer.records.append((2.0 * M * kk, e0, e1, "f16s1_bneck", 4.0 * (M * (bk.cin + 256) + kk)))
    return out


def split_planes_f16x2(w):
    """[2, *w.shape] fp16 planes (w1 = fp16(w), w2 = fp16((w - w1) * 2048)) of an fp32 tensor (lvc_split_weights)."""
    _req_cuda(w)
    w = w.detach().float().contiguous()
    out = torch.empty((2,) + tuple(w.shape), device=w.device, dtype=torch.float16)
    check(_lib.lib().lvc_split_weights(ptr(w), c_longlong(w.numel()), c_int(2), ptr(out), ptr(_conv_error_view(w.device)), _stream(w)),
          "lvc_split_weights")
    return out


def stem_conv_pool(x4, pc, relu=True, second=None):
    """x4 [N,H,W,4] NHWC4, pc = pack_conv(stem weight 64x3x7x7, bn=..., stride=2, pad=3, stem=True) ->
    [N,Hp,Wp,64]: conv 7x7/2 + affine + ReLU + max-pool 3x3/2 in one launch."""
    _req_cuda(x4)
    assert x4.dim() == 4 and x4.shape[3] == 4 and x4.is_contiguous() and x4.dtype == torch.float32
    assert pc.mode == 1 and pc.K == 64 and pc.R == 7 and pc.stride == 2 and pc.pad == 3
    N, H, W, _ = x4.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
    out = torch.empty(N, Hp, Wp, 64, device=x4.device, dtype=torch.float32)
    y2 = second(out.shape) if second is not None else None     # [N,Hp,Wp,64] view with a wider row stride
    if y2 is not None:
        assert y2.shape == out.shape and y2.stride(3) == 1 and y2.stride(1) == Wp * y2.stride(2) and y2.stride(0) == Hp * y2.stride(1)
    if CONV_SPLIT == "f16x2":
        ws = conv_workspace(x4.device)
        err = ws[1024 * 256 * 128 * 4 + 1024 * 4: 1024 * 256 * 128 * 4 + 1024 * 4 + 4]   # the conv error word
        st = _lib.lib().lvc_stem_conv_pool_nhwc4_f16x2(ptr(x4), ptr(pc.split2h()), ptr(pc.scale), ptr(pc.shift), ptr(out),
                                                       c_int(N), c_int(H), c_int(W), c_int(pc.w.shape[0]),
                                                       c_int(1 if relu else 0), ptr(err), ptr(y2),
                                                       c_int(y2.stride(2) if y2 is not None else 0), _stream(x4))
        check(st, "lvc_stem_conv_pool_nhwc4_f16x2")
        return out
    raise RuntimeError("the fused stem exists for the fp16 split only; BasicStem runs conv + max-pool as two launches otherwise")


def linear(x, pc, relu=False, split=None):
    """x: [M, K_in] -> [M, K_out] through the same MFMA kernel.  split: as conv2d_nhwc (gradient GEMMs pass
    DGRAD_SPLIT: unscaled gradients sit below fp16's normal range, and the fp16 split's range word only sees overflow)."""
    M, Kin = x.shape
    y = conv2d_nhwc(x.view(M, 1, 1, Kin), pc, relu=relu, split=split)
    return y.view(M, pc.K)


# --------------------------------------------------------------------------- ROIAlign
def new_status(device):
    return torch.zeros(1, dtype=torch.int32, device=device)


def roi_align_forward(input, rois, spatial_scale, pooled_h, pooled_w, sampling_ratio, aligned, status=None):
    """Reference-shaped op (csrc/vision.cpp:96): NCHW input, rois [K,5], returns [K,C,ph,pw]."""
    _req_cuda(input, rois)
    input = input.contiguous().float()
    rois = rois.contiguous().float()
    B, C, H, W = input.shape
    K = rois.shape[0]
    out = torch.empty(K, C, pooled_h, pooled_w, device=input.device, dtype=torch.float32)
    st = _lib.lib().lvc_roi_align_forward_nchw(
        ptr(input), ptr(rois), ptr(out), c_int(B), c_int(C), c_int(H), c_int(W), c_int(K),
        c_int(pooled_h), c_int(pooled_w), c_float(spatial_scale), c_int(sampling_ratio),
        c_int(1 if aligned else 0), ptr(status), _stream(input))
    check(st, "lvc_roi_align_forward_nchw")
    return out


ROI_ORDER_MIN = 1024     # RoIs per launch from which a work order is computed (lvc_roi_work_order: one small launch)


def roi_align_fpn_nhwc(feats, scales, rois, levels, pooled_h, pooled_w, sampling_ratio, aligned,
                       num_valid=None, status=None, order=None):
    """feats: list of NHWC level tensors [B,H_l,W_l,C]; rois [K,5]; levels [K] int32 (or None when one
    level).  Returns [K, ph, pw, C]."""
    _req_cuda(rois, *feats)
    L = len(feats)
    B, _, _, C = feats[0].shape
    K = rois.shape[0]
    rois = rois.contiguous().float()
    out = torch.empty(K, pooled_h, pooled_w, C, device=rois.device, dtype=torch.float32)
    FP = c_void_p * L
    IP = c_int * L
    FL = c_float * L
    fp = FP(*[f.data_ptr() for f in feats])
    hs = IP(*[f.shape[1] for f in feats])
    ws = IP(*[f.shape[2] for f in feats])
    sc = FL(*[float(s) for s in scales])
    for f in feats:
        assert f.is_contiguous() and f.dtype == torch.float32 and f.shape[0] == B and f.shape[3] == C
    if levels is not None:
        assert levels.dtype == torch.int32 and levels.is_contiguous()
    if order is not None:       # a caller's own work order ([K] int32 permutation; experiments)
        assert order.dtype == torch.int32 and order.is_contiguous() and order.numel() == K
    elif K >= ROI_ORDER_MIN:
        # the workgroups take the RoIs largest window first (the launch's time follows the window area; in proposal order a few
        # large RoIs start last and finish alone)
        order = torch.empty(K, device=rois.device, dtype=torch.int32)
        if B <= 16:
            # image b's RoIs on XCD b % 8, inside an image by (level, band of rows): what an XCD reads close in time shares its L2
            check(_lib.lib().lvc_roi_work_order_xcd(ptr(rois), ptr(levels), c_int(K), c_int(B), ptr(order), _stream(rois)), "lvc_roi_work_order_xcd")
        else:
            check(_lib.lib().lvc_roi_work_order(ptr(rois), ptr(levels), sc, c_int(L), c_int(K), c_int(pooled_h), ptr(order), _stream(rois)),
                  "lvc_roi_work_order")
    st = _lib.lib().lvc_roi_align_fpn_nhwc_ordered(
        fp, hs, ws, sc, c_int(L), c_int(B), c_int(C), ptr(rois), ptr(levels), ptr(num_valid), c_int(K),
        c_int(pooled_h), c_int(pooled_w), c_int(sampling_ratio), c_int(1 if aligned else 0), ptr(out),
        ptr(status), ptr(order), _stream(rois))
    check(st, "lvc_roi_align_fpn_nhwc")
    return out


def roi_align_backward(grad, rois, spatial_scale, pooled_h, pooled_w, batch_size, channels, height, width,
                       sampling_ratio, aligned, status=None):
    """Reference-shaped op (csrc/vision.cpp:97): grad [K,C,ph,pw], rois [K,5] -> grad_input [B,C,H,W]."""
    _req_cuda(grad, rois)
    grad = grad.contiguous().float()
    rois = rois.contiguous().float()
    K = rois.shape[0]
    assert grad.shape == (K, channels, pooled_h, pooled_w)
    gin = torch.empty(batch_size, channels, height, width, device=grad.device, dtype=torch.float32)
    st = _lib.lib().lvc_roi_align_backward_nchw(
        ptr(grad), ptr(rois), ptr(gin), c_int(batch_size), c_int(channels), c_int(height), c_int(width), c_int(K),
        c_int(pooled_h), c_int(pooled_w), c_float(spatial_scale), c_int(sampling_ratio),
        c_int(1 if aligned else 0), ptr(status), _stream(grad))
    check(st, "lvc_roi_align_backward_nchw")
    return gin


def roi_align_fpn_backward_nhwc(grad, shapes, scales, rois, levels, sampling_ratio, aligned, num_valid=None,
                                status=None):
    """Gradient of roi_align_fpn_nhwc.  grad [K,ph,pw,C]; shapes: list of (B,H_l,W_l,C).  Returns the list of
    NHWC level gradients."""
    _req_cuda(grad, rois)
    grad = grad.contiguous().float()
    rois = rois.contiguous().float()
    K, ph, pw, C = grad.shape
    L = len(shapes)
    B = shapes[0][0]
    outs = [torch.empty(*sh, device=grad.device, dtype=torch.float32) for sh in shapes]
    FP = c_void_p * L
    IP = c_int * L
    FL = c_float * L
    if levels is not None:
        assert levels.dtype == torch.int32 and levels.is_contiguous()
    st = _lib.lib().lvc_roi_align_fpn_backward_nhwc(
        ptr(grad), FP(*[o.data_ptr() for o in outs]), IP(*[sh[1] for sh in shapes]), IP(*[sh[2] for sh in shapes]),
        FL(*[float(s) for s in scales]), c_int(L), c_int(B), c_int(C), ptr(rois), ptr(levels), ptr(num_valid),
        c_int(K), c_int(ph), c_int(pw), c_int(sampling_ratio), c_int(1 if aligned else 0), ptr(status),
        _stream(grad))
    check(st, "lvc_roi_align_fpn_backward_nhwc")
    return outs


# --------------------------------------------------------------------------- NMS
def batched_nms_batch(boxes, scores, idxs, counts, iou_threshold, max_keep=0):
    """boxes [B,Nmax,4], scores [B,Nmax], idxs [B,Nmax] int32 or None, counts [B] int32 device or None.
    Returns (keep [B,Nmax] int32, num_keep [B] int32), all on device, no sync."""
    _req_cuda(boxes, scores, idxs, counts)
    B, Nmax = scores.shape
    boxes = boxes.contiguous().float()
    scores = scores.contiguous().float()
    if idxs is not None:
        idxs = idxs.contiguous()
        assert idxs.dtype == torch.int32
    keep = torch.empty(B, Nmax, dtype=torch.int32, device=boxes.device)
    num_keep = torch.empty(B, dtype=torch.int32, device=boxes.device)
    wsb = _lib.lib().lvc_batched_nms_workspace_bytes(c_int(B), c_int(Nmax))
    ws = torch.empty(wsb, dtype=torch.uint8, device=boxes.device)
    st = _lib.lib().lvc_batched_nms(
        ptr(boxes), ptr(scores), ptr(idxs), ptr(counts), c_int(B), c_int(Nmax), c_double(iou_threshold),
        c_int(max_keep), ptr(keep), ptr(num_keep), ptr(ws), c_longlong(wsb), _stream(boxes))
    check(st, "lvc_batched_nms")
    return keep, num_keep


def batched_nms(boxes, scores, idxs, iou_threshold):
    """Drop-in for reference detectron2/layers/nms.py:10-29 (single image): int64 keep indices,
    score-descending."""
    n = boxes.shape[0]
    if n == 0:
        return torch.empty(0, dtype=torch.int64, device=boxes.device)
    idx32 = idxs.to(torch.int32)[None] if idxs is not None else None
    keep, nk = batched_nms_batch(boxes[None], scores[None], idx32, None, iou_threshold)
    return keep[0, : int(nk.item())].to(torch.int64)


def nms(boxes, scores, iou_threshold):
    """Drop-in for torchvision.ops.nms as imported by reference detectron2/layers/nms.py:7."""
    return batched_nms(boxes, scores, None, iou_threshold)


# --------------------------------------------------------------------------- box pipeline
SCALE_CLAMP = 4.135166556742356  # log(1000/16), reference detectron2/modeling/box_regression.py:11-12


def _arr(ctype, vals):
    return (ctype * len(vals))(*vals)


def rpn_proposals(logits, deltas, cell_anchors, strides, image_sizes, pre_nms_topk, post_nms_topk,
                  nms_thresh, min_box_size=0.0):
    """find_top_rpn_proposals on device (reference proposal_utils.py:13-118 + rpn.py:489-508).

    logits[l]: [B,H,W,ld] view whose channel a holds the objectness of anchor a (any channel stride);
    deltas[l]: [B,H,W,ld'] view whose channel a*4+c holds delta c of anchor a.  Both may be channel
    slices of one fused NHWC tensor (pass tensor[..., :A] and tensor[..., A:]).
    cell_anchors[l]: [A,4] device tensors; image_sizes: [B,2] int32 device (h, w).
    Returns (boxes [B,post,4], objectness_logits [B,post], count [B] int32); rows past count are zero.
    """
    L = len(logits)
    B, A = logits[0].shape[0], cell_anchors[0].shape[0]
    dev = logits[0].device
    _req_cuda(*logits, *deltas, *cell_anchors, image_sizes)
    for t in list(logits) + list(deltas):
        assert t.dtype == torch.float32 and t.stride(-1) == 1 and t.dim() == 4
        assert t.stride(2) == t.stride(3) * 0 + t.stride(2)  # pixel-major view
    Hs = [t.shape[1] for t in logits]
    Ws = [t.shape[2] for t in logits]
    for t, h, w in zip(list(logits) + list(deltas), Hs * 2, Ws * 2):
        # rows must be dense pixels: stride(1) == W*stride(2), stride(0) == H*W*stride(2)
        assert t.stride(1) == w * t.stride(2) and t.stride(0) == h * w * t.stride(2), "need an NHWC channel slice"
    VP, IP = c_void_p * L, c_int * L
    lp = VP(*[t.data_ptr() for t in logits])
    dp = VP(*[t.data_ptr() for t in deltas])
    ap = VP(*[t.contiguous().data_ptr() for t in cell_anchors])
    ldl = IP(*[t.stride(2) for t in logits])
    ldd = IP(*[t.stride(2) for t in deltas])
    hs, ws_, st = IP(*Hs), IP(*Ws), IP(*strides)
    lib = _lib.lib()
    lib.lvc_rpn_proposals_workspace_bytes.restype = c_longlong
    nbytes = lib.lvc_rpn_proposals_workspace_bytes(c_int(B), c_int(L), c_int(A), hs, ws_, c_int(pre_nms_topk))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    boxes = torch.empty(B, post_nms_topk, 4, device=dev, dtype=torch.float32)
    olog = torch.empty(B, post_nms_topk, device=dev, dtype=torch.float32)
    count = torch.empty(B, dtype=torch.int32, device=dev)
    assert image_sizes.dtype == torch.int32 and image_sizes.is_contiguous()
    rc = lib.lvc_rpn_proposals(lp, ldl, dp, ldd, ap, hs, ws_, st, c_int(L), c_int(A), c_int(B), ptr(image_sizes),
                               c_int(pre_nms_topk), c_int(post_nms_topk), c_double(nms_thresh),
                               c_float(min_box_size), c_float(SCALE_CLAMP), ptr(boxes), ptr(olog), ptr(count),
                               ptr(ws), c_longlong(nbytes), _stream(boxes))
    check(rc, "lvc_rpn_proposals")
    return boxes, olog, count


def assign_levels_rois(boxes, min_level, max_level, canonical_box_size=224, canonical_level=4):
    """boxes [B,R,4] -> (levels [B*R] int32, rois [B*R,5]) (reference poolers.py:23-59, 69-96)."""
    _req_cuda(boxes)
    B, R, _ = boxes.shape
    boxes = boxes.contiguous()
    levels = torch.empty(B * R, dtype=torch.int32, device=boxes.device)
    rois = torch.empty(B * R, 5, dtype=torch.float32, device=boxes.device)
    rc = _lib.lib().lvc_assign_levels_rois(ptr(boxes), c_int(B), c_int(R), c_int(min_level), c_int(max_level),
                                           c_int(canonical_box_size), c_int(canonical_level), ptr(levels), ptr(rois),
                                           _stream(boxes))
    check(rc, "lvc_assign_levels_rois")
    return levels, rois


def fast_rcnn_inference(cls_logits, deltas, proposals, prop_count, image_sizes, num_classes, box_weights,
                        score_thresh, nms_thresh, topk, post=None, status=None, max_candidates=16384):
    """predict_boxes/predict_probs + fast_rcnn_inference (+ detector_postprocess when `post` is given).

    cls_logits [B*R, >=K+1], deltas [B*R, 4K or 4] (row-contiguous, any row stride), proposals [B,R,4],
    prop_count [B] int32 or None, image_sizes [B,2] int32, post [B,4] fp32 = (scale_x, scale_y, out_h, out_w).
    Returns (boxes [B,topk,4], scores [B,topk], classes [B,topk] int32, rows [B,topk] int32, count [B] int32).
    max_candidates: capacity of the per-image (roi, class) candidate list; None = R*K (cannot overflow).  The default
    keeps the NMS in its one-block form; an overflow sets status bit 1 (value 2) -> `CandidateOverflow` where the status
    is read, and the model entry points re-run with the full capacity.
    """
    _req_cuda(cls_logits, deltas, proposals, prop_count, image_sizes, post)
    B, R, _ = proposals.shape
    K = num_classes
    dev = proposals.device
    assert cls_logits.stride(1) == 1 and deltas.stride(1) == 1
    cls_agnostic = 1 if deltas.shape[1] == 4 else 0
    if status is None:
        status = new_status(dev)
    lib = _lib.lib()
    lib.lvc_fast_rcnn_inference_workspace_bytes.restype = c_longlong
    max_candidates = max(1, R * K) if max_candidates is None else min(max_candidates, max(1, R * K))
    nbytes = lib.lvc_fast_rcnn_inference_workspace_bytes(c_int(B), c_int(max_candidates))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ob = torch.empty(B, topk, 4, device=dev, dtype=torch.float32)
    osc = torch.empty(B, topk, device=dev, dtype=torch.float32)
    ocl = torch.empty(B, topk, device=dev, dtype=torch.int32)
    orow = torch.empty(B, topk, device=dev, dtype=torch.int32)
    cnt = torch.empty(B, device=dev, dtype=torch.int32)
    wx, wy, ww, wh = box_weights
    rc = lib.lvc_fast_rcnn_inference(
        ptr(cls_logits), c_int(cls_logits.stride(0)), ptr(deltas), c_int(deltas.stride(0)), c_int(K),
        c_int(cls_agnostic), ptr(proposals.contiguous()), ptr(prop_count), c_int(B), c_int(R), ptr(image_sizes),
        c_float(wx), c_float(wy), c_float(ww), c_float(wh), c_float(SCALE_CLAMP), c_float(score_thresh),
        c_double(nms_thresh), c_int(topk), c_int(max_candidates), ptr(post), ptr(ob), ptr(osc), ptr(ocl), ptr(orow),
        ptr(cnt), ptr(status), ptr(ws), c_longlong(nbytes), _stream(proposals))
    check(rc, "lvc_fast_rcnn_inference")
    return ob, osc, ocl, orow, cnt


# --------------------------------------------------------------------------- trunk elementwise
def preprocess_into(image, out_slot, mean, std):
    """image: CHW (3,h,w) float32 or uint8 device tensor; out_slot: [Hp,Wp,4] view of the batch tensor.
    out = (image - mean) / std, zero padded, 4th channel zero (reference rcnn.py:324-333, image_list.py:95-119)."""
    _req_cuda(image, out_slot)
    assert image.dim() == 3 and image.shape[0] == 3
    if image.dtype == torch.uint8:
        dt = 1
    else:
        image = image.float()
        dt = 0
    image = image.contiguous()
    Hp, Wp, four = out_slot.shape
    assert four == 4 and out_slot.is_contiguous()
    m = (c_float * 3)(*[float(v) for v in mean])
    s = (c_float * 3)(*[float(v) for v in std])
    rc = _lib.lib().lvc_preprocess_nhwc4(ptr(image), c_int(dt), c_int(image.shape[1]), c_int(image.shape[2]), m, s,
                                         ptr(out_slot), c_int(Hp), c_int(Wp), _stream(image))
    check(rc, "lvc_preprocess_nhwc4")


def preprocess_batch_into(images, out, mean, std):
    """images: list of CHW (3,h,w) device tensors (all float32 or all uint8); out: [B,Hp,Wp,4] batch tensor.  One launch per
    16 images (lvc_preprocess_batch_nhwc4); the same values as `preprocess_into` image by image."""
    _req_cuda(out, *images)
    B = len(images)
    assert out.dim() == 4 and out.shape[0] == B and out.shape[3] == 4 and out.is_contiguous() and out.dtype == torch.float32
    u8 = images[0].dtype == torch.uint8
    imgs = []
    for im in images:
        assert im.dim() == 3 and im.shape[0] == 3
        if u8:
            assert im.dtype == torch.uint8
        else:
            im = im.float()
        imgs.append(im.contiguous())
    P = (c_void_p * B)(*[im.data_ptr() for im in imgs])
    hs = (c_int * B)(*[int(im.shape[1]) for im in imgs])
    ws = (c_int * B)(*[int(im.shape[2]) for im in imgs])
    m = (c_float * 3)(*[float(v) for v in mean])
    s_ = (c_float * 3)(*[float(v) for v in std])
    rc = _lib.lib().lvc_preprocess_batch_nhwc4(P, c_int(1 if u8 else 0), hs, ws, c_int(B), m, s_, ptr(out), c_int(out.shape[1]),
                                               c_int(out.shape[2]), _stream(out))
    check(rc, "lvc_preprocess_batch_nhwc4")
    return imgs   # keeps converted copies alive until the caller drops them (the launch is asynchronous)


def resize_bilinear_u8(img, new_h, new_w, coeffs_fn, out_slot=None, mean=None, std=None):
    """Pillow-exact bilinear resize of a uint8 [H,W,3] image on the device (csrc/resize.hip).  coeffs_fn(in, out,
    device) -> (bounds, coefficients, ksize) (lvc_amd.data.transforms.resample_coeffs).  Returns the uint8
    [new_h,new_w,3] result; with out_slot [Hp,Wp,4] fp32 also writes the normalised, zero-padded NHWC4 pixels."""
    if not img.is_cuda:
        raise RuntimeError("resize_bilinear_u8 needs a device tensor (move the uint8 image first: 1 byte per sample)")
    img = img.contiguous()
    H, W, _ = img.shape
    dev = img.device
    xb = xk = yb = yk = tmp = None
    kxs = kys = 0
    if new_w != W:
        xb, xk, kxs = coeffs_fn(W, new_w, dev)
        tmp = torch.empty(H * new_w * 3, dtype=torch.uint8, device=dev)
    if new_h != H:
        yb, yk, kys = coeffs_fn(H, new_h, dev)
    out = torch.empty(new_h, new_w, 3, dtype=torch.uint8, device=dev)
    Hp = Wp = 0
    m = s = None
    if out_slot is not None:
        Hp, Wp, four = out_slot.shape
        assert four == 4 and out_slot.is_contiguous() and out_slot.dtype == torch.float32
        m = (c_float * 3)(*[float(v) for v in mean])
        s = (c_float * 3)(*[float(v) for v in std])
    rc = _lib.lib().lvc_resize_bilinear_u8(ptr(img), c_int(H), c_int(W), c_int(new_h), c_int(new_w), ptr(xb), ptr(xk),
                                           c_int(kxs), ptr(yb), ptr(yk), c_int(kys), ptr(tmp), ptr(out), ptr(out_slot),
                                           c_int(Hp), c_int(Wp), m, s, _stream(img))
    check(rc, "lvc_resize_bilinear_u8")
    return out


def maxpool2d_nhwc(x, k, stride, pad):
    _req_cuda(x)
    assert x.is_contiguous() and x.dtype == torch.float32
    N, H, W, C = x.shape
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    y = torch.empty(N, Ho, Wo, C, device=x.device, dtype=torch.float32)
    rc = _lib.lib().lvc_maxpool2d_nhwc(ptr(x), ptr(y), c_int(N), c_int(H), c_int(W), c_int(C), c_int(k), c_int(stride),
                                       c_int(pad), _stream(x))
    check(rc, "lvc_maxpool2d_nhwc")
    return y


def rownorm(x, mu=None, eps=1e-5, mode=0, out=None):
    """y = (x - mu) / (|x - mu| + eps)  (mode 0)   or   / max(|x - mu|, eps)  (mode 1), row-wise."""
    _req_cuda(x, mu)
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32
    M, D = x.shape
    if out is None:
        out = torch.empty(M, D, device=x.device, dtype=torch.float32)
    rc = _lib.lib().lvc_rownorm(ptr(x), ptr(mu), ptr(out), c_int(M), c_int(D), c_int(x.stride(0)), c_int(out.stride(0)),
                                c_float(eps), c_int(mode), _stream(x))
    check(rc, "lvc_rownorm")
    return out


def rownorm_backward(x, dy, eps=1e-5, accumulate_into=None):
    """Gradient of `rownorm(x, eps=eps, mode=0)` w.r.t. x; accumulate_into: a [M,D] tensor the result is added to."""
    _req_cuda(x, dy)
    x, dy = x.contiguous(), dy.contiguous()
    M, D = x.shape
    dx = accumulate_into if accumulate_into is not None else torch.empty_like(x)
    assert dx.is_contiguous() and dx.shape == x.shape
    check(_lib.lib().lvc_rownorm_backward(ptr(x), ptr(dy), ptr(dx), c_int(M), c_int(D), c_float(eps),
                                          c_int(1 if accumulate_into is not None else 0), _stream(x)), "lvc_rownorm_backward")
    return dx


# --------------------------------------------------------------------------- label-verification kNN
def colmean(x):
    _req_cuda(x)
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32
    mu = torch.empty(x.shape[1], device=x.device, dtype=torch.float32)
    rc = _lib.lib().lvc_colmean(ptr(x), ptr(mu), c_int(x.shape[0]), c_int(x.shape[1]), c_int(x.stride(0)), _stream(x))
    check(rc, "lvc_colmean")
    return mu


def knn_topk_vote(sims, num_shots, shot_classes, det_classes, k):
    """sims [Q, >=S] fp32; shot_classes [S] int64; det_classes [Q] int64 or None.
    Returns (top10 class ids [Q,10] int64, keep [Q] int64 or None)."""
    _req_cuda(sims, shot_classes, det_classes)
    Q = sims.shape[0]
    assert sims.stride(1) == 1 and shot_classes.dtype == torch.int64 and shot_classes.is_contiguous()
    top = torch.empty(Q, 10, dtype=torch.int64, device=sims.device)
    keep = torch.empty(Q, dtype=torch.int64, device=sims.device) if det_classes is not None else None
    if det_classes is not None:
        det_classes = det_classes.contiguous()
        assert det_classes.dtype == torch.int64
    rc = _lib.lib().lvc_knn_topk_vote(ptr(sims), c_int(sims.stride(0)), c_int(Q), c_int(num_shots), ptr(shot_classes),
                                      ptr(det_classes), c_int(k), ptr(top), ptr(keep), _stream(sims))
    check(rc, "lvc_knn_topk_vote")
    return top, keep


def knn_topk_vote_blocks(sims, num_shots, shot_classes, det_classes, k, block=4096):
    """`knn_topk_vote` for any number of shots (LVIS-sized sets): the columns are cut into equal blocks of <= 4096 shots, each
    block's ten best (similarity, shot index) pairs per row are found (lvc_knn_topk_candidates), and one merge launch ranks
    the lists, gathers the classes and votes (lvc_knn_merge_vote).  The exact top ten of a row are among the per-block top
    tens, and both steps use the reference's tie rule (lower shot index first): the same answer as one launch over the row."""
    _req_cuda(sims, shot_classes, det_classes)
    Q = sims.shape[0]
    assert sims.stride(1) == 1 and shot_classes.dtype == torch.int64 and shot_classes.is_contiguous()
    nblk = (num_shots + block - 1) // block
    assert 1 <= nblk <= 64, "at most 64 x 4096 shots"
    per = (num_shots + nblk - 1) // nblk
    cv = torch.empty(nblk, Q, 10, dtype=torch.float32, device=sims.device)
    ci = torch.empty(nblk, Q, 10, dtype=torch.int32, device=sims.device)
    lib = _lib.lib()
    for b in range(nblk):
        s0, s1 = b * per, min(num_shots, (b + 1) * per)
        check(lib.lvc_knn_topk_candidates(ptr(sims[:, s0:]), c_int(sims.stride(0)), c_int(Q), c_int(s1 - s0), c_int(s0), ptr(cv[b]),
                                          ptr(ci[b]), _stream(sims)), "lvc_knn_topk_candidates")
    top = torch.empty(Q, 10, dtype=torch.int64, device=sims.device)
    keep = torch.empty(Q, dtype=torch.int64, device=sims.device) if det_classes is not None else None
    if det_classes is not None:
        det_classes = det_classes.contiguous()
        assert det_classes.dtype == torch.int64
    check(lib.lvc_knn_merge_vote(ptr(cv), ptr(ci), c_int(nblk), c_int(Q), ptr(shot_classes), ptr(det_classes), c_int(k), ptr(top),
                                 ptr(keep), _stream(sims)), "lvc_knn_merge_vote")
    return top, keep


def max_f32(x):
    """[1] fp32 device tensor: the maximum of x (lvc_max_f32)."""
    _req_cuda(x)
    x = x.contiguous()
    assert x.dtype == torch.float32
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    check(_lib.lib().lvc_max_f32(ptr(x), c_longlong(x.numel()), ptr(out), _stream(x)), "lvc_max_f32")
    return out


def knn_margins(qres, sres_max, acc, extra=0.0):
    """The two-stage sweep's per-row margins (label_verification.pre_filter_margins + extra) in one launch (lvc_knn_margins).
    qres [Q] fp32, sres_max [1] fp32 device tensor."""
    _req_cuda(qres, sres_max)
    qres = qres.contiguous()
    assert qres.dtype == torch.float32 and sres_max.dtype == torch.float32 and sres_max.numel() == 1
    out = torch.empty_like(qres)
    check(_lib.lib().lvc_knn_margins(ptr(qres), ptr(sres_max), c_float(acc), c_float(extra), c_int(qres.numel()), ptr(out), _stream(qres)),
          "lvc_knn_margins")
    return out


def rownorm_h(x, mu=None, eps=1e-5, mode=0, want_rows=True, want_resid=False):
    """`rownorm` for the two-stage kNN sweep: (y fp32 [M,D] or None, yh fp16 [M,D], den [M]); y is bit-identical to
    `rownorm`'s output and equals (x - mu) / den[:, None] exactly.  want_resid: also resid [M] = |row - fp16(row)|_2."""
    _req_cuda(x, mu)
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32
    M, D = x.shape
    y = torch.empty(M, D, device=x.device, dtype=torch.float32) if want_rows else None
    yh = torch.empty(M, D, device=x.device, dtype=torch.float16)
    den = torch.empty(M, device=x.device, dtype=torch.float32)
    resid = torch.empty(M, device=x.device, dtype=torch.float32) if want_resid else None
    rc = _lib.lib().lvc_rownorm_h(ptr(x), ptr(mu), ptr(y), ptr(yh), ptr(den), ptr(resid), c_int(M), c_int(D), c_int(x.stride(0)),
                                  c_float(eps), c_int(mode), _stream(x))
    check(rc, "lvc_rownorm_h")
    return (y, yh, den, resid) if want_resid else (y, yh, den)


def gemm_f16(a, b, n=None, ldb=None, q15=False):
    """a [M,C] . b^T on fp16 operands with fp32 accumulation -> [M,N] fp32 (lvc_gemm_f16): the pre-filter of the two-stage
    kNN sweep.  n / ldb: use n rows of b that lie ldb elements apart (a strided subset of a contiguous [*, C] tensor).
    q15: the result as 16-bit fixed point, rint(32766 * value), NaN -> 32767 (lvc_gemm_f16_q15; rows padded to a multiple of 8 columns,
    returned as the [M, N] view)."""
    _req_cuda(a, b)
    M, C = a.shape
    assert a.is_contiguous() and b.is_contiguous() and a.dtype == torch.float16 and b.dtype == torch.float16 and b.shape[1] == C
    N = b.shape[0] if n is None else n
    ldb = C if ldb is None else ldb
    assert (N - 1) * ldb + C <= b.numel()
    if q15:
        ld = (N + 7) // 8 * 8
        y = torch.empty(M, ld, device=a.device, dtype=torch.int16)
        rc = _lib.lib().lvc_gemm_f16_q15(ptr(a), ptr(b), c_int(ldb), ptr(y), c_int(M), c_int(N), c_int(C), c_int(ld), _stream(a))
        check(rc, "lvc_gemm_f16_q15")
        return y[:, :N]
    y = torch.empty(M, N, device=a.device, dtype=torch.float32)
    rc = _lib.lib().lvc_gemm_f16(ptr(a), ptr(b), c_int(ldb), ptr(y), c_int(M), c_int(N), c_int(C), c_int(N), _stream(a))
    check(rc, "lvc_gemm_f16")
    return y


KNN_LIST_CAP = 256


def knn_verify_topk_vote(approx, q, sn, margin, shot_classes, det_classes, k, mu=None, den=None, margins=None):
    """approx [Q,S] from gemm_f16 over the fp16 roundings of the normalised rows; q [Q,D] the raw query descriptors with the
    mu / den `rownorm_h` used (or the normalised rows themselves with mu = den = None); sn [S,D] normalised shots.
    Exact top-10 classes + vote (csrc/knn.hip)."""
    _req_cuda(approx, q, sn, shot_classes, det_classes, mu, den, margins)
    Q, S = approx.shape[0], sn.shape[0]
    assert margins is None or (margins.dtype == torch.float32 and margins.is_contiguous() and margins.numel() == Q)
    assert approx.stride(1) == 1 and q.stride(1) == 1 and sn.is_contiguous() and shot_classes.dtype == torch.int64
    assert q.dtype == torch.float32 and sn.dtype == torch.float32 and q.shape[1] == sn.shape[1]
    top = torch.empty(Q, 10, dtype=torch.int64, device=approx.device)
    keep = torch.empty(Q, dtype=torch.int64, device=approx.device) if det_classes is not None else None
    if det_classes is not None:
        det_classes = det_classes.contiguous()
        assert det_classes.dtype == torch.int64
    fn = _lib.lib().lvc_knn_verify_topk_vote_q15 if approx.dtype == torch.int16 else _lib.lib().lvc_knn_verify_topk_vote
    assert approx.dtype in (torch.int16, torch.float32)
    rc = fn(ptr(approx), c_int(approx.stride(0)), c_int(Q), c_int(S), ptr(q), c_int(q.stride(0)),
            ptr(mu), ptr(den), ptr(sn), c_int(sn.shape[1]), c_float(margin), ptr(margins),
            ptr(shot_classes), ptr(det_classes), c_int(k), ptr(top), ptr(keep), _stream(approx))
    check(rc, "lvc_knn_verify_topk_vote")
    return top, keep


# --------------------------------------------------------------------------- training-time kernels
def match_boxes(gt_boxes, boxes, thresholds, labels, allow_low_quality_matches):
    """pairwise_iou + Matcher on device.  gt_boxes [G,4] (G >= 1), boxes [N,4].
    thresholds: 1 or 2 floats, labels: len(thresholds)+1 ints.  Returns (matches int64 [N], labels int8 [N], vals [N])."""
    _req_cuda(gt_boxes, boxes)
    G, N = gt_boxes.shape[0], boxes.shape[0]
    gt_boxes = gt_boxes.contiguous().float()
    boxes = boxes.contiguous().float()
    dev = boxes.device
    matches = torch.empty(N, dtype=torch.int64, device=dev)
    mlabels = torch.empty(N, dtype=torch.int8, device=dev)
    vals = torch.empty(N, dtype=torch.float32, device=dev)
    scratch = torch.empty(max(G, 1), dtype=torch.int32, device=dev)
    nthr = len(thresholds)
    assert nthr in (1, 2) and len(labels) == nthr + 1
    t = list(thresholds) + [0.0]
    lab = list(labels) + [0]
    rc = _lib.lib().lvc_match_boxes(ptr(gt_boxes), c_int(G), ptr(boxes), c_int(N), c_float(t[0]), c_float(t[1]),
                                    c_int(nthr), c_int(lab[0]), c_int(lab[1]), c_int(lab[2]),
                                    c_int(1 if allow_low_quality_matches else 0), ptr(matches), ptr(mlabels), ptr(vals),
                                    ptr(scratch), _stream(boxes))
    check(rc, "lvc_match_boxes")
    return matches, mlabels, vals


def cat_ground_truth(gt_instances):
    """The images' gt boxes one after the other + the prefix of their counts as a device int32 [B+1] (pinned staging, asynchronous copy:
    a plain torch.tensor(..., device=) would wait for everything queued on the stream -- the trunk).  -> (gt [G,4], gt_off, counts list)."""
    dev = gt_instances[0].gt_boxes.tensor.device
    lens = [len(t) for t in gt_instances]
    off = [0]
    for n in lens:
        off.append(off[-1] + n)
    gt = torch.cat([t.gt_boxes.tensor for t in gt_instances], 0).float().contiguous() if off[-1] else torch.zeros(0, 4, device=dev)
    gt_off = torch.tensor(off, dtype=torch.int32).pin_memory().to(dev, non_blocking=True)
    return gt, gt_off, lens


def match_boxes_batched(gt, gt_off, B, boxes, nbox, thresholds, labels, allow_low_quality_matches, return_vals=False):
    """pairwise_iou + Matcher for B images in two launches (csrc/train_targets.hip).  gt [G,4] / gt_off int32 [B+1] from
    `cat_ground_truth`; boxes [N,4] shared by the images (anchors) or [B,N,4] with nbox int32 [B] rows in use per image (None: all).
    -> (matches int32 [B,N], labels int8 [B,N])."""
    _req_cuda(gt_off, boxes)
    shared = boxes.dim() == 2
    N = boxes.shape[-2]
    boxes = boxes.contiguous()
    assert boxes.dtype == torch.float32 and gt.dtype == torch.float32 and gt_off.dtype == torch.int32
    dev = boxes.device
    matches = torch.empty(B, N, dtype=torch.int32, device=dev)
    mlabels = torch.empty(B, N, dtype=torch.int8, device=dev)
    vals = torch.empty(B, N, dtype=torch.float32, device=dev)
    G = gt.shape[0]
    scratch = torch.empty(max(G, 1), dtype=torch.int32, device=dev)
    nthr = len(thresholds)
    assert nthr in (1, 2) and len(labels) == nthr + 1
    t = list(thresholds) + [0.0]
    lab = list(labels) + [0]
    rc = _lib.lib().lvc_match_boxes_batched(ptr(gt.contiguous()), ptr(gt_off), c_int(G), c_int(B), ptr(boxes), c_longlong(0 if shared else N * 4),
                                            ptr(nbox), c_int(N), c_float(t[0]), c_float(t[1]), c_int(nthr), c_int(lab[0]), c_int(lab[1]),
                                            c_int(lab[2]), c_int(1 if allow_low_quality_matches else 0), ptr(matches), ptr(mlabels), ptr(vals),
                                            ptr(scratch), _stream(boxes))
    check(rc, "lvc_match_boxes_batched")
    if return_vals:
        return matches, mlabels, vals       # vals: the best IoU of each box (Matcher's matched_vals)
    return matches, mlabels


_SUBSAMPLE_WS = {}


def randperm_is_patched():
    """True when torch.randperm has been replaced (the parity tests substitute arange to obtain the reference's deterministic choice):
    the samplers then take their keys from it instead of generating them in the kernel."""
    return getattr(torch.randperm, "__name__", "") != "randperm"


def sampling_keys(B, N, device):
    """(keys, seed) for `subsample_batched`: normally (None, a 62-bit seed from torch's CPU generator -- torch.manual_seed governs it, no
    device work); with a patched torch.randperm its B * N values as int64 [B,N]."""
    if randperm_is_patched():
        return torch.randperm(B * N, device=device).view(B, N), 0
    return None, int(torch.randint(0, 1 << 62, (1,)).item())


def subsample_batched(labels, keys, cap_pos, bs, seed=0):
    """subsample_labels (reference sampling.py:10-54) for every row of labels int8 [B,N] (1 positive, 0 negative, else ignored) in one
    launch: the min(#pos, cap_pos) positives and min(#neg, bs - num_pos) negatives with the smallest keys (int64 [B,N], distinct, e.g.
    torch.randperm(B * N); None: generated in the kernel from `seed`, see `sampling_keys`).  -> (sel int32 [B,bs]: positives first, each group by increasing key, -1 padded; counts int32 [B,2])."""
    _req_cuda(labels)
    B, N = labels.shape
    assert labels.dtype == torch.int8 and (keys is None or (keys.is_cuda and keys.dtype == torch.int64 and keys.shape == labels.shape))
    sel = torch.empty(B, bs, dtype=torch.int32, device=labels.device)
    counts = torch.empty(B, 2, dtype=torch.int32, device=labels.device)
    nbits = max(1, int(B * N - 1).bit_length())
    lib = _lib.lib()
    lib.lvc_subsample_workspace_bytes.restype = c_longlong
    key = (labels.device.index, torch.cuda.current_stream(labels.device).cuda_stream, B)
    ws = _SUBSAMPLE_WS.get(key)
    if ws is None:      # zeroed once; the kernels leave it zeroed (per stream: two launches in flight must not share the histograms)
        if len(_SUBSAMPLE_WS) > 16:
            _SUBSAMPLE_WS.clear()
        ws = _SUBSAMPLE_WS[key] = torch.zeros(lib.lvc_subsample_workspace_bytes(c_int(B)), dtype=torch.uint8, device=labels.device)
    from ctypes import c_ulonglong

    rc = lib.lvc_subsample_batched(ptr(labels.contiguous()), ptr(keys.contiguous() if keys is not None else None), c_ulonglong(int(seed)),
                                   c_int(B), c_int(N), c_int(nbits), c_int(cap_pos), c_int(bs), ptr(sel), ptr(counts), ptr(ws), _stream(labels))
    check(rc, "lvc_subsample_batched")
    return sel, counts


def rpn_gather_sampled(fused, A, cell_anchors, strides, sel, counts, matches, gt, gt_off):
    """Rows of the sampled anchors straight from the head's per-level outputs fused[l] [B,H,W,ld] (channel a = objectness, A + 4 a + c =
    delta c).  -> (logits [S], deltas [S,4], anchors [S,4], matched gt boxes [S,4], labels int8 [S]) with S = B * bs; padding rows carry
    label -1."""
    L = len(fused)
    B, bs = sel.shape
    dev = sel.device
    for t in fused:
        assert t.dim() == 4 and t.dtype == torch.float32 and t.stride(3) == 1 and t.stride(1) == t.shape[2] * t.stride(2) and t.stride(0) == t.shape[1] * t.shape[2] * t.stride(2)
    VP, IP = c_void_p * L, c_int * L
    S = B * bs
    logits = torch.empty(S, device=dev)
    deltas = torch.empty(S, 4, device=dev)
    anchors = torch.empty(S, 4, device=dev)
    gtb = torch.empty(S, 4, device=dev)
    labels = torch.empty(S, dtype=torch.int8, device=dev)
    cells = [c.contiguous() for c in cell_anchors]
    rc = _lib.lib().lvc_rpn_gather_sampled(VP(*[t.data_ptr() for t in fused]), IP(*[t.stride(2) for t in fused]), VP(*[c.data_ptr() for c in cells]),
                                           IP(*[t.shape[1] for t in fused]), IP(*[t.shape[2] for t in fused]), IP(*[int(s) for s in strides]),
                                           c_int(L), c_int(A), c_int(B), c_int(bs), ptr(sel), ptr(counts), ptr(matches), ptr(gt), ptr(gt_off),
                                           ptr(logits), ptr(deltas), ptr(anchors), ptr(gtb), ptr(labels), _stream(sel))
    check(rc, "lvc_rpn_gather_sampled")
    return logits, deltas, anchors, gtb, labels


def roi_build_table(pboxes, plogits, pcount, gt, gt_off, gt_logit, Wt):
    """proposals [B,P,4] / [B,P] / count int32 [B] + gt -> (boxes [B,Wt,4], logits [B,Wt], rows in use int32 [B]): every image's proposals
    followed by its gt boxes (add_ground_truth_to_proposals)."""
    _req_cuda(pboxes, plogits, pcount, gt_off)
    B, P = plogits.shape
    dev = pboxes.device
    boxes = torch.empty(B, Wt, 4, device=dev)
    logits = torch.empty(B, Wt, device=dev)
    nrow = torch.empty(B, dtype=torch.int32, device=dev)
    rc = _lib.lib().lvc_roi_build_table(ptr(pboxes.contiguous()), ptr(plogits.contiguous()), ptr(pcount), c_int(B), c_int(P), ptr(gt), ptr(gt_off),
                                        c_float(gt_logit), c_int(Wt), ptr(boxes), ptr(logits), ptr(nrow), _stream(pboxes))
    check(rc, "lvc_roi_build_table")
    return boxes, logits, nrow


def roi_gather_sampled(boxes, logits, matches, sel, counts, gt_classes, gt_off, num_classes):
    """Sampled rows of the table -> (boxes [B,bs,4], logits [B,bs], classes int64 [B,bs] (K = background, -1 padding), matched gt index
    int64 [B,bs])."""
    B, Wt = logits.shape
    bs = sel.shape[1]
    dev = boxes.device
    sb = torch.empty(B, bs, 4, device=dev)
    sl = torch.empty(B, bs, device=dev)
    sc = torch.empty(B, bs, dtype=torch.int64, device=dev)
    sm = torch.empty(B, bs, dtype=torch.int64, device=dev)
    assert gt_classes.dtype == torch.int64
    rc = _lib.lib().lvc_roi_gather_sampled(ptr(boxes), ptr(logits), ptr(matches), ptr(sel), ptr(counts), ptr(gt_classes.contiguous()), ptr(gt_off),
                                           c_int(B), c_int(Wt), c_int(bs), c_int(num_classes), ptr(sb), ptr(sl), ptr(sc), ptr(sm), _stream(boxes))
    check(rc, "lvc_roi_gather_sampled")
    return sb, sl, sc, sm


def fast_rcnn_losses(logits, deltas, proposals, gt_boxes, gt_classes, num_classes, box_weights, smooth_l1_beta):
    """Returns (losses [2] = (loss_cls, loss_box_reg), dlogits [R,K+1], ddeltas [R,4K|4])."""
    _req_cuda(logits, deltas, proposals, gt_boxes, gt_classes)
    R = logits.shape[0]
    K = num_classes
    nreg = deltas.shape[1] if deltas.shape[1] == 4 else 4 * K
    dev = logits.device
    out = torch.empty(2, device=dev, dtype=torch.float32)
    dl = torch.empty(R, K + 1, device=dev, dtype=torch.float32)
    dd = torch.empty(R, nreg, device=dev, dtype=torch.float32)
    row_terms = torch.empty(2 * R, device=dev, dtype=torch.float64)
    wx, wy, ww, wh = box_weights
    assert logits.stride(1) == 1 and deltas.stride(1) == 1 and gt_classes.dtype == torch.int64
    rc = _lib.lib().lvc_fast_rcnn_losses(ptr(logits), c_int(logits.stride(0)), ptr(deltas), c_int(deltas.stride(0)), c_int(K),
                                         c_int(1 if nreg == 4 else 0), ptr(proposals.contiguous()), ptr(gt_boxes.contiguous()),
                                         ptr(gt_classes.contiguous()), c_int(R), c_float(wx), c_float(wy), c_float(ww),
                                         c_float(wh), c_float(smooth_l1_beta), ptr(out), ptr(dl), ptr(dd), ptr(row_terms),
                                         _stream(logits))
    check(rc, "lvc_fast_rcnn_losses")
    return out, dl, dd


def giou_box_loss(deltas, proposals, gt_boxes, gt_classes, num_classes, box_weights, scale_clamp, iterate=False,
                  lambda_=0.0):
    """Box-corrector regression loss.  Returns (loss [1], ddeltas [R,4])."""
    _req_cuda(deltas, proposals, gt_boxes, gt_classes)
    R = deltas.shape[0]
    out = torch.empty(1, device=deltas.device, dtype=torch.float32)
    dd = torch.empty(R, 4, device=deltas.device, dtype=torch.float32)
    wx, wy, ww, wh = box_weights
    assert deltas.stride(1) == 1 and gt_classes.dtype == torch.int64
    rc = _lib.lib().lvc_giou_box_loss(ptr(deltas), c_int(deltas.stride(0)), ptr(proposals.contiguous()),
                                      ptr(gt_boxes.contiguous()), ptr(gt_classes.contiguous()), c_int(R),
                                      c_int(num_classes), c_float(wx), c_float(wy), c_float(ww), c_float(wh),
                                      c_float(scale_clamp), c_int(1 if iterate else 0), c_float(lambda_), ptr(out),
                                      ptr(dd), _stream(deltas))
    check(rc, "lvc_giou_box_loss")
    return out, dd


def relu_backward(dy, y):
    _req_cuda(dy, y)
    dy, y = dy.contiguous(), y.contiguous()
    out = torch.empty_like(dy)
    check(_lib.lib().lvc_relu_backward(ptr(dy), ptr(y), c_longlong(dy.numel()), ptr(out), _stream(dy)), "lvc_relu_backward")
    return out


def colsum(x):
    """x [M,N] row-major -> [N] column sums (rows added in order)."""
    _req_cuda(x)
    x = x.contiguous()
    M, N = x.shape
    out = torch.empty(N, device=x.device, dtype=torch.float32)
    check(_lib.lib().lvc_colsum(ptr(x), c_int(M), c_int(N), c_int(N), ptr(out), _stream(x)), "lvc_colsum")
    return out


def colsum_rows(x2d):
    """x2d [M,N] row-major with M up to 10^5-10^6 (conv bias gradients): slab sums + fp32 atomics."""
    _req_cuda(x2d)
    x2d = x2d.contiguous()
    M, N = x2d.shape
    out = torch.empty(N, device=x2d.device, dtype=torch.float32)
    check(_lib.lib().lvc_colsum_atomic(ptr(x2d), c_int(M), c_int(N), c_int(N), ptr(out), _stream(x2d)), "lvc_colsum_atomic")
    return out


BWD_TIMER = None     # a list: every weight- / data-gradient launch appends (kind, engine, flops, algorithmic bytes, start, end events)


def _bwd_timed(kind, engine, flops, nbytes, fn, tag=""):
    if BWD_TIMER is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    BWD_TIMER.append((kind, engine, flops, nbytes, e0, e1, tag))
    return out


def conv_wgrad(x, dy, scale, R, S, stride, pad, split=None):
    """dW of y = conv(x, W) (* scale per output channel): x [N,H,W,C], dy [N,Ho,Wo,K] -> [K,R,S,C] fp32."""
    if BWD_TIMER is not None:
        eng = "f16x2" if (split or DGRAD_SPLIT) == "f16x2" else WGRAD_ENGINE
        fl = 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * dy.shape[3] * x.shape[3] * R * S
        nb = 4.0 * (x.numel() + dy.numel() + dy.shape[3] * R * S * x.shape[3])
        tag = "%dx%dx%d %d>%d %dx%d s%d" % (dy.shape[0], dy.shape[1], dy.shape[2], x.shape[3], dy.shape[3], R, S, stride)
        return _bwd_timed("wgrad", eng, fl, nb, lambda: _conv_wgrad(x, dy, scale, R, S, stride, pad, split), tag)
    return _conv_wgrad(x, dy, scale, R, S, stride, pad, split)


def _conv_wgrad(x, dy, scale, R, S, stride, pad, split=None):
    _req_cuda(x, dy, scale)
    assert x.dim() == 4 and dy.dim() == 4 and x.is_contiguous() and dy.is_contiguous()
    assert x.dtype == torch.float32 and dy.dtype == torch.float32
    N, H, W, C = x.shape
    K = dy.shape[3]
    assert dy.shape[0] == N and dy.shape[1] == (H + 2 * pad - R) // stride + 1 and dy.shape[2] == (W + 2 * pad - S) // stride + 1
    dw = torch.empty(K, R, S, C, device=x.device, dtype=torch.float32)
    if (split or DGRAD_SPLIT) == "f16x2":   # gradients scaled into fp16's range (solver.LossScaler): fp16 MFMA path
        check(_lib.lib().lvc_conv_wgrad_nhwc_f16x2(ptr(x), ptr(dy), ptr(scale), ptr(dw), c_int(N), c_int(H), c_int(W),
                                                   c_int(C), c_int(K), c_int(R), c_int(S), c_int(stride), c_int(pad),
                                                   c_int(K), ptr(_conv_error_view(x.device)), _stream(x)),
              "lvc_conv_wgrad_nhwc_f16x2")
        return dw
    fn = "lvc_conv_wgrad_nhwc_bf16x3" if WGRAD_ENGINE == "bf16x3" else "lvc_conv_wgrad_nhwc"
    check(getattr(_lib.lib(), fn)(ptr(x), ptr(dy), ptr(scale), ptr(dw), c_int(N), c_int(H), c_int(W), c_int(C), c_int(K),
                                  c_int(R), c_int(S), c_int(stride), c_int(pad), c_int(K), _stream(x)), fn)
    return dw


# ---- deferred, grouped weight gradients (csrc/conv_wgrad.hip: conv_wgrad_group_bf16x3_kernel) --------------------------------------
DEFER_WGRAD = _os.environ.get("LVC_DEFER_WGRAD", "1") != "0"
_WGRAD_GROUP = 24            # jobs per launch (kernel-argument table)
_WGRAD_Q = []                # (param, x, g, scale, R, stride, pad)
_WGRAD_ARMED = [False]
_WGRAD_SINKS = {}            # id(param) -> (destination(param) -> tensor the gradient is written into, ready(param))
# "1": the grouped launches on a second stream.  Measured on the box-corrector step (2 images per GPU): 28.41 / 28.46 / 28.49 ms against
# 28.28 / 28.68 / 29.24 on the pass's own stream -- the step is bound by the host's launch rate there, not by the chip; off by default
WGRAD_SIDE_STREAM = _os.environ.get("LVC_WGRAD_SIDE_STREAM", "0") != "0"
_WGRAD_SIDE = {}             # device index -> the stream the grouped launches run on
_WGRAD_PENDING = []          # (event, deliveries, operands kept alive) of groups in flight on the side stream


_WGRAD_USES = {}             # id(param) -> differentiable forward uses whose backward node has not run yet (layers.wrappers._ConvFn)


def register_wgrad_sink(param, destination, ready, owner=None):
    """A gradient exchange that owns flat buckets hands out the bucket slice as the place the deferred weight gradient is written
    to (no copy into the bucket afterwards) and is told when it is there (`lvc_amd.distributed.GradientBuckets`).  The sink dies with
    the exchange: bound methods are held through weak references (and `owner`, if given, through another) -- a dropped exchange must
    not keep receiving gradients into its dead buckets, nor be kept alive by this table.  One live sink per parameter."""
    import weakref

    def weak(fn):
        try:
            return weakref.WeakMethod(fn)
        except TypeError:          # a plain function: nothing to die with
            return lambda: fn

    if _wgrad_sink(param) is not None:
        raise RuntimeError("lvc_amd: parameter already has a deferred-gradient sink (a second GradientBuckets over the same parameters? "
                           "call remove() on the first)")
    _WGRAD_SINKS[id(param)] = (weak(destination), weak(ready), weakref.ref(param), weakref.ref(owner) if owner is not None else None)


def unregister_wgrad_sink(param):
    _WGRAD_SINKS.pop(id(param), None)


def _wgrad_sink(param):
    """(destination, ready) of the live sink of `param`, or None (and the entry is dropped) when its parameter, its owner or its
    callables are gone -- ids are reused."""
    sink = _WGRAD_SINKS.get(id(param))
    if sink is None:
        return None
    dst, rdy = sink[0](), sink[1]()
    if sink[2]() is not param or (sink[3] is not None and sink[3]() is None) or dst is None or rdy is None:
        del _WGRAD_SINKS[id(param)]
        return None
    return dst, rdy


def reset_wgrad_queue():
    del _WGRAD_Q[:]
    del _WGRAD_PENDING[:]
    _WGRAD_USES.clear()
    _WGRAD_ARMED[0] = False


def in_backward():
    """True while the autograd engine is executing a graph task on this thread."""
    return torch._C._current_graph_task_id() != -1


def note_wgrad_use(param):
    """A differentiable forward use of `param` (layers.wrappers._ConvFn.forward): its backward node will deliver one weight gradient."""
    _WGRAD_USES[id(param)] = _WGRAD_USES.get(id(param), 0) + 1


def wgrad_use_done(param):
    """That use's backward node ran (it queued a deferred job or handed its gradient to AccumulateGrad)."""
    n = _WGRAD_USES.get(id(param), 0)
    if n > 0:
        _WGRAD_USES[id(param)] = n - 1


def wgrad_queued(param):
    """True while a deferred weight gradient of `param` waits in the queue or more uses of it have yet to run their backward:
    its gradient is not final (lvc_amd.distributed.GradientBuckets._hook)."""
    return _WGRAD_USES.get(id(param), 0) > 0 or any(j[0] is param for j in _WGRAD_Q)


def can_defer_wgrad(x, g):
    """Deferred grouped launch: the default unscaled bf16x3 path inside a `backward()` (not torch.autograd.grad: the gradient is
    written to `param.grad` by a callback at the end of the pass, as AccumulateGrad would)."""
    return (DEFER_WGRAD and WGRAD_ENGINE == "bf16x3" and DGRAD_SPLIT != "f16x2" and x.numel() * 4 < (1 << 31) and g.numel() * 4 < (1 << 31)
            and x.is_contiguous() and g.is_contiguous())


def defer_wgrad(param, x, g, scale, R, stride, pad):
    """Queue dW of y = conv(x, param) (* scale) for the grouped launch; `param.grad` holds it when `backward()` returns."""
    _WGRAD_Q.append((param, x, g, scale, R, stride, pad))
    if not _WGRAD_ARMED[0]:
        _WGRAD_ARMED[0] = True
        torch.autograd.Variable._execution_engine.queue_callback(_flush_wgrad_end)
    if sum(1 for j in _WGRAD_Q if _WGRAD_USES.get(id(j[0]), 0) == 0) >= _WGRAD_GROUP:
        flush_wgrad()


def _flush_wgrad_end():
    _WGRAD_ARMED[0] = False
    flush_wgrad(final=True)


def flush_wgrad(final=False):
    """Launch the queued weight gradients: one zeroing, one grouped wgrad launch, one grouped layout / accumulate launch."""
    if final:
        q = list(_WGRAD_Q)
        del _WGRAD_Q[:]
        # uses counted by a forward whose backward never ran (two forwards, one backward) made `wgrad_queued` hold back parameters that
        # got their whole gradient through AccumulateGrad: report those now, the pass is over
        stale = [i for i, n in _WGRAD_USES.items() if n > 0]
        _WGRAD_USES.clear()
        queued = {id(j[0]) for j in q}
        for i in stale:
            entry = _WGRAD_SINKS.get(i)
            if entry is not None and i not in queued:
                prm = entry[2]()
                sink = _wgrad_sink(prm) if prm is not None else None
                if sink is not None and prm.grad is not None:
                    sink[1](prm)
    else:
        # a flush DELIVERS its parameters (p.grad is set, a bucketed exchange may start reducing them): only parameters whose every use
        # of this pass has run its backward node go out -- the RPN head's convolutions are used once per pyramid level, and a weight
        # whose uses straddled two flushes was accumulated into a bucket slice that was already being reduced
        q = [j for j in _WGRAD_Q if _WGRAD_USES.get(id(j[0]), 0) == 0]
        _WGRAD_Q[:] = [j for j in _WGRAD_Q if _WGRAD_USES.get(id(j[0]), 0) != 0]
    if not q:
        if final:
            _retire_wgrad()
        return
    import ctypes

    dev = q[0][1].device
    n = len(q)
    first = {}               # a parameter used several times in the pass (the RPN head over the pyramid levels): ONE accumulation buffer
    offs, total = [], 0
    for j, (p, *_r) in enumerate(q):
        if id(p) in first:
            offs.append(offs[first[id(p)]])
        else:
            first[id(p)] = j
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
    flat = torch.zeros(total, device=dev, dtype=torch.float32)
    PA = ctypes.c_void_p * n
    xs, dys, scs, dws = PA(), PA(), PA(), PA()
    nf = len(first)
    PF = ctypes.c_void_p * nf
    srcs, dsts = PF(), PF()
    shapes = (c_int * (10 * n))()
    fshapes = (c_int * (4 * nf))()
    outs = []
    flops = nbytes = 0.0
    f = 0
    for j, (p, x, g, scale, R, stride, pad) in enumerate(q):
        N, H, W, C = x.shape
        Kc = g.shape[3]
        xs[j], dys[j], scs[j] = x.data_ptr(), g.data_ptr(), (scale.data_ptr() if scale is not None else None)
        dws[j] = flat.data_ptr() + 4 * offs[j]
        shapes[10 * j: 10 * j + 10] = [N, H, W, C, Kc, R, R, stride, pad, Kc]
        flops += 2.0 * g.shape[0] * g.shape[1] * g.shape[2] * Kc * C * R * R
        nbytes += 4.0 * (x.numel() + g.numel() + p.numel())
        if first[id(p)] != j:
            continue
        sink = _wgrad_sink(p)
        if p.grad is not None:
            dst, beta = p.grad, 1
            assert dst.is_contiguous() and dst.dtype == torch.float32
        else:
            dst = sink[0](p) if sink is not None else torch.empty_like(p, memory_format=torch.contiguous_format)
            beta = 0
        outs.append((p, dst, beta, sink))
        srcs[f], dsts[f] = dws[j], dst.data_ptr()
        fshapes[4 * f: 4 * f + 4] = [Kc, C, R * R, beta]
        f += 1
    main = torch.cuda.current_stream(dev)
    side = main
    if WGRAD_SIDE_STREAM and BWD_TIMER is None:
        # off the critical path in time as well: the group runs on a second stream next to the data-gradient chain, whose launches
        # (8 400-pixel maps at 2 images per GPU) leave most of the chip idle
        side = _WGRAD_SIDE.get(dev.index)
        if side is None:
            side = _WGRAD_SIDE[dev.index] = torch.cuda.Stream(dev)
        side.wait_stream(main)              # x, dy, the zeroed buffer and the destinations are complete on the main stream
    st = c_void_p(side.cuda_stream)

    def launch():
        check(_lib.lib().lvc_conv_wgrad_group_bf16x3(c_int(n), xs, dys, scs, dws, shapes, st), "lvc_conv_wgrad_group_bf16x3")

    _bwd_timed("wgrad", "bf16x3", flops, nbytes, launch, "group of %d" % n)
    check(_lib.lib().lvc_wgrad_finalize_group(c_int(nf), srcs, dsts, fshapes, st), "lvc_wgrad_finalize_group")
    if side is main:
        _deliver_wgrad(outs)
        return
    ev = torch.cuda.Event()
    ev.record(side)
    _WGRAD_PENDING.append((ev, outs, (q, flat)))     # operands stay alive until the main stream has passed the group's event
    _retire_wgrad(keep=0 if final else 1)


def _deliver_wgrad(outs):
    for p, dst, beta, sink in outs:
        if not beta:
            p.grad = dst
        if sink is not None:
            sink[1](p)


def _retire_wgrad(keep=0):
    """Hand the finished groups' gradients over (main stream ordered behind them); the newest `keep` groups stay in flight."""
    while len(_WGRAD_PENDING) > keep:
        ev, outs, _alive = _WGRAD_PENDING.pop(0)
        torch.cuda.current_stream().wait_event(ev)
        _deliver_wgrad(outs)


def scatter_stride2(x, H, W):
    """x [N,(H-1)//2+1,(W-1)//2+1,C] -> [N,H,W,C] with x at the even pixels and zeros elsewhere."""
    _req_cuda(x)
    x = x.contiguous()
    N, Hs, Ws, C = x.shape
    assert Hs == (H - 1) // 2 + 1 and Ws == (W - 1) // 2 + 1
    y = torch.empty(N, H, W, C, device=x.device, dtype=torch.float32)
    check(_lib.lib().lvc_scatter_stride2_nhwc(ptr(x), ptr(y), c_int(N), c_int(H), c_int(W), c_int(C), _stream(x)),
          "lvc_scatter_stride2_nhwc")
    return y


def subsample2_into(x, out):
    """out[n, i, j, :C] = x[n, 2i, 2j, :] where out is a [N,Hs,Ws,C] view with unit channel stride (rows may be wider: the tail
    channels of a concat buffer)."""
    _req_cuda(x, out)
    N, H, W, C = x.shape
    Hs, Ws = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    assert x.is_contiguous() and x.dtype == torch.float32 and out.dtype == torch.float32 and out.shape == (N, Hs, Ws, C)
    ld = out.stride(2)
    assert out.stride(3) == 1 and out.stride(1) == Ws * ld and out.stride(0) == Hs * Ws * ld
    check(_lib.lib().lvc_subsample2_nhwc(ptr(x), ptr(out), c_int(N), c_int(H), c_int(W), c_int(C), c_int(ld), _stream(x)),
          "lvc_subsample2_nhwc")
    return out


def downsum2x2(x):
    """x [N,2Hs,2Ws,C] -> [N,Hs,Ws,C] sums of the 2x2 blocks."""
    _req_cuda(x)
    x = x.contiguous()
    N, H, W, C = x.shape
    assert H % 2 == 0 and W % 2 == 0
    y = torch.empty(N, H // 2, W // 2, C, device=x.device, dtype=torch.float32)
    check(_lib.lib().lvc_downsum2x2_nhwc(ptr(x), ptr(y), c_int(N), c_int(H // 2), c_int(W // 2), c_int(C), _stream(x)),
          "lvc_downsum2x2_nhwc")
    return y


def pack_conv_dgrad(weight, scale, pad):
    """Packed weights of the DATA gradient of y = conv(x, weight, stride 1 after sub-sampling, pad) * scale:
    dx = conv(dy, Wt, pad = R-1-pad) with Wt[c][k][r][s] = scale[k] * weight[k][c][R-1-r][S-1-s].  A strided 1x1 is
    the same product on the sub-sampled grid followed by `scatter_stride2`.  Output channels of the forward conv
    (the contraction here) are zero-padded to the kernels' 32-channel chunk."""
    Kout, C, R, S = weight.shape
    kin_pad = (Kout + 31) // 32 * 32
    rows_pad = (C + BN - 1) // BN * BN
    planes = _planes_hint(R, S, kin_pad, DGRAD_SPLIT)
    wp, pl = _pack_weights(weight, scale, rows_pad, kin_pad, 1, planes)
    return _with_planes(PackedConv(wp, None, None, C, kin_pad, R, S, 1, R - 1 - pad, R * S * kin_pad, 0), planes, pl)


def conv_dgrad(dy, pcd, x_shape, stride):
    """dx [x_shape] of a conv whose `pack_conv_dgrad` is pcd.  dy [N,Ho,Wo,K] contiguous."""
    N, H, W, C = x_shape
    if dy.shape[3] != pcd.C:   # contraction padded to 32 channels
        dy = torch.nn.functional.pad(dy, (0, pcd.C - dy.shape[3]))
    if BWD_TIMER is not None:
        fl = 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * dy.shape[3] * pcd.K * pcd.R * pcd.S
        nb = 4.0 * (dy.numel() + dy.shape[0] * dy.shape[1] * dy.shape[2] * pcd.K + pcd.K * pcd.R * pcd.S * dy.shape[3])
        tag = "%dx%dx%d %d>%d %dx%d" % (dy.shape[0], dy.shape[1], dy.shape[2], dy.shape[3], pcd.K, pcd.R, pcd.S)
        dxs = _bwd_timed("dgrad", DGRAD_SPLIT, fl, nb, lambda: conv2d_nhwc(dy.contiguous(), pcd, split=DGRAD_SPLIT), tag)
    else:
        dxs = conv2d_nhwc(dy.contiguous(), pcd, split=DGRAD_SPLIT)
    if stride == 1:
        assert tuple(dxs.shape) == (N, H, W, C), (dxs.shape, x_shape)
        return dxs
    assert stride == 2 and pcd.R == 1, "strided 3x3 convolutions are not on the path (STRIDE_IN_1X1)"
    return scatter_stride2(dxs, H, W)


def linear_backward(x, weight, dz, need_dx=True, need_dw=True):
    """Gradients of y = x @ weight.T on the conv/GEMM kernel.  x [M,Kin], weight [Kout,Kin], dz [M,Kout] (already
    masked by the activation).  Returns (dx [M,Kin] or None, dw [Kout,Kin] or None).  Both products contract over a
    dimension that is zero-padded to the kernel's 32-wide k chunk (Kout for dx, M for dw)."""
    M, Kin = x.shape
    Kout = weight.shape[0]
    dev = x.device
    dx = dw = None
    if need_dx:   # dx = dz @ W: contraction over Kout, "weights" = W^T [Kin rows, Kout]
        pk = (-Kout) % 32
        wt = torch.zeros(Kin, Kout + pk, device=dev)
        wt[:, :Kout] = weight.detach().t()
        dzp = dz
        if pk:
            dzp = torch.zeros(M, Kout + pk, device=dev)
            dzp[:, :Kout] = dz
        dx = linear(dzp.contiguous(), pack_linear(wt, split=DGRAD_SPLIT), split=DGRAD_SPLIT)
    if need_dw and Kout % 4 == 0 and Kin % 4 == 0:   # dw = dz^T @ x on the pixel-contraction kernel (no transposes)
        dw = conv_wgrad(x.detach().contiguous().view(M, 1, 1, Kin), dz.contiguous().view(M, 1, 1, Kout), None, 1, 1, 1, 0)
        dw = dw.view(Kout, Kin)
    elif need_dw:   # dw = dz^T @ x: contraction over M, "weights" = x^T [Kin rows, M]
        pm = (-M) % 32
        xt = torch.zeros(Kin, M + pm, device=dev)
        xt[:, :M] = x.detach().t()
        dzt = torch.zeros(Kout, M + pm, device=dev)
        dzt[:, :M] = dz.t()
        dw = linear(dzt, pack_linear(xt, split=DGRAD_SPLIT), split=DGRAD_SPLIT)
    return dx, dw


def rpn_losses(logits, deltas, anchors, gt_boxes, labels, smooth_l1_beta, normalizer, with_grad=False):
    """Sampled-anchor RPN losses.  All inputs are rows gathered at the sampled anchors.  with_grad: also returns
    d(loss_cls)/d(logits) [S] and d(loss_loc)/d(deltas) [S,4]."""
    _req_cuda(logits, deltas, anchors, gt_boxes, labels)
    S = logits.shape[0]
    out = torch.zeros(2, device=logits.device, dtype=torch.float32)
    assert labels.dtype == torch.int8
    if with_grad:
        dl = torch.empty(S, device=logits.device, dtype=torch.float32)
        dd = torch.empty(S, 4, device=logits.device, dtype=torch.float32)
        rc = _lib.lib().lvc_rpn_losses_grad(ptr(logits.contiguous()), ptr(deltas.contiguous()), ptr(anchors.contiguous()),
                                            ptr(gt_boxes.contiguous()), ptr(labels.contiguous()), c_int(S),
                                            c_float(smooth_l1_beta), c_float(normalizer), ptr(out), ptr(dl), ptr(dd),
                                            _stream(logits))
        check(rc, "lvc_rpn_losses_grad")
        return out, dl, dd
    rc = _lib.lib().lvc_rpn_losses(ptr(logits.contiguous()), ptr(deltas.contiguous()), ptr(anchors.contiguous()),
                                   ptr(gt_boxes.contiguous()), ptr(labels.contiguous()), c_int(S), c_float(smooth_l1_beta),
                                   c_float(normalizer), ptr(out), _stream(logits))
    check(rc, "lvc_rpn_losses")
    return out


def decode_boxes(deltas, boxes, weights, image_sizes=None):
    """Class-agnostic apply_deltas (+ clip to the row's image): deltas [B*R, >=4], boxes [B,R,4] -> [B,R,4]."""
    _req_cuda(deltas, boxes, image_sizes)
    B, R, _ = boxes.shape
    boxes = boxes.contiguous()
    out = torch.empty_like(boxes)
    assert deltas.stride(1) == 1 and deltas.shape[0] == B * R
    wx, wy, ww, wh = weights
    rc = _lib.lib().lvc_decode_boxes(ptr(deltas), c_int(deltas.stride(0)), ptr(boxes), c_int(B * R), c_int(R), ptr(image_sizes),
                                     c_float(wx), c_float(wy), c_float(ww), c_float(wh), c_float(SCALE_CLAMP), ptr(out),
                                     _stream(boxes))
    check(rc, "lvc_decode_boxes")
    return out


def crop_resize_nearest(image, windows, out_size=224):
    """image [C,H,W] fp32 device; windows [K,8] int32 device (x1,y1,x2,y2,l_pad,t_pad,side_w,side_h) -> [K,C,out,out]."""
    _req_cuda(image, windows)
    C, H, W = image.shape
    Kn = windows.shape[0]
    image = image.contiguous().float()
    assert windows.dtype == torch.int32 and windows.is_contiguous()
    out = torch.empty(Kn, C, out_size, out_size, device=image.device, dtype=torch.float32)
    rc = _lib.lib().lvc_crop_resize_nearest(ptr(image), c_int(C), c_int(H), c_int(W), ptr(windows), c_int(Kn), c_int(out_size),
                                            ptr(out), _stream(image))
    check(rc, "lvc_crop_resize_nearest")
    return out


# --------------------------------------------------------------------------- descriptor network (ViT) pieces
def vit_patchify(img, patch_size, kpad=None, mean=None, std=None):
    """img [B,C,H,W] fp32 -> [B*P, kpad] rows (column c*ps*ps + r*ps + s; zero columns up to kpad, a multiple of 32).
    mean / std (sequences of C floats): the rows of (img - mean[c]) / std[c] -- get_descriptors' crop normalisation fused in."""
    _req_cuda(img)
    B, C, H, W = img.shape
    kc = C * patch_size * patch_size
    kpad = kpad or kc
    P = (H // patch_size) * (W // patch_size)
    if kpad == kc:
        out = torch.empty(B * P, kc, device=img.device, dtype=torch.float32)
        if mean is not None:
            m = (c_float * C)(*[float(v) for v in mean])
            s_ = (c_float * C)(*[float(v) for v in std])
            check(_lib.lib().lvc_vit_patchify_norm(ptr(img), m, s_, ptr(out), c_int(B), c_int(C), c_int(H), c_int(W), c_int(patch_size),
                                                   _stream(img)), "lvc_vit_patchify_norm")
            return out
        check(_lib.lib().lvc_vit_patchify(ptr(img), ptr(out), c_int(B), c_int(C), c_int(H), c_int(W), c_int(patch_size), _stream(img)),
              "lvc_vit_patchify")
        return out
    tmp = vit_patchify(img, patch_size, mean=mean, std=std)
    out = torch.zeros(B * P, kpad, device=img.device, dtype=torch.float32)
    out[:, :kc] = tmp
    return out


def vit_tokens(emb, cls, pos, B):
    _req_cuda(emb, cls, pos)
    D = emb.shape[1]
    P = emb.shape[0] // B
    out = torch.empty(B * (P + 1), D, device=emb.device, dtype=torch.float32)
    check(_lib.lib().lvc_vit_tokens(ptr(emb.contiguous()), ptr(cls.detach().contiguous()), ptr(pos.detach().contiguous()), ptr(out),
                                    c_int(B), c_int(P), c_int(D), _stream(emb)), "lvc_vit_tokens")
    return out


def layernorm(x, weight, bias, eps):
    _req_cuda(x, weight, bias)
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32
    M, D = x.shape
    y = torch.empty(M, D, device=x.device, dtype=torch.float32)
    check(_lib.lib().lvc_layernorm(ptr(x), c_int(x.stride(0)), ptr(weight.detach()), ptr(bias.detach()), ptr(y), c_int(D), c_int(M),
                                   c_int(D), c_float(eps), _stream(x)), "lvc_layernorm")
    return y


def gelu(x):
    _req_cuda(x)
    x = x.contiguous()
    y = torch.empty_like(x)
    check(_lib.lib().lvc_gelu(ptr(x), ptr(y), c_longlong(x.numel()), _stream(x)), "lvc_gelu")
    return y


MHA_MFMA = True   # 0: the one-thread-per-query fp32 VALU kernel of round 2 (lvc_mha)


QKV_PLANES = _os.environ.get("LVC_QKV_PLANES", "1") != "0"
_MHA_PLANES_WS = {}


def can_qkv_planes(x, pc, num_heads, head_dim):
    """The qkv GEMM may write the attention's fp16 operand planes from its epilogue (csrc/conv_pw_s1.hip PLANES instance): the layer
    runs on the single-accumulator pointwise kernel and the attention on the matrix-core kernel."""
    return (QKV_PLANES and MHA_MFMA and CONV_ENGINE == "bf16x3" and CONV_SPLIT == "f16x2" and PW_S1 == 2 and head_dim == 64
            and not pc.two_acc and pc.state.get("tier", 0) == 0 and pc.R == 1 and pc.stride == 1 and pc.C >= _PW_S1_ONE_MIN_C and pc.C % 32 == 0
            and pc.K == 3 * num_heads * 64 and x.numel() * 4 < (1 << 31)
            and x.shape[0] >= 2048 and x.shape[0] * pc.K < (1 << 29))      # the rows for which `conv2d_nhwc` takes that kernel too


def qkv_attention(x, pc, B, N, num_heads, scale):
    """softmax(q k^T scale) v of a ViT block from the block's normalised input x [B*N, C] and its packed qkv layer: the qkv GEMM writes
    q * scale * log2(e), k, v as fp16 (hi, lo) planes from its epilogue (lvc_conv1x1_qkv_planes_f16s1: the fp32 qkv tensor and the
    split pass of lvc_mha_mfma do not exist), lvc_mha_mfma_planes consumes them.  Bit-identical to `mha(linear(x, pc), ...)`.
    -> [B*N, H*64]."""
    _req_cuda(x)
    x = x.contiguous()
    dev = x.device
    lib = _lib.lib()
    lib.lvc_mha_workspace_bytes.restype = c_longlong
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, B, N, num_heads)
    ws = _MHA_PLANES_WS.get(key)
    if ws is None:      # zeroed ONCE: the epilogue never writes the padding rows N..Npad-1, which must stay zero
        if len(_MHA_PLANES_WS) > 8:
            _MHA_PLANES_WS.clear()
        ws = _MHA_PLANES_WS[key] = torch.zeros(max(16, lib.lvc_mha_workspace_bytes(c_int(B), c_int(N), c_int(num_heads))), dtype=torch.uint8, device=dev)
    planes, scale2 = pc.split2s()
    pc.last_one = True
    lib.lvc_set_range_slot(c_int(pc.slot))
    try:
        check(lib.lvc_conv1x1_qkv_planes_f16s1(ptr(x), ptr(planes), ptr(scale2), ptr(pc.shift), ptr(ws), c_int(B), c_int(N), c_int(pc.C),
                                               c_int(num_heads), c_float(scale), ptr(_conv_error_view(dev)),
                                               ptr(conv_workspace(dev)), _stream(x)), "lvc_conv1x1_qkv_planes_f16s1")
    finally:
        lib.lvc_set_range_slot(c_int(0))
    out = torch.empty(B * N, num_heads * 64, device=dev, dtype=torch.float32)
    check(lib.lvc_mha_mfma_planes(ptr(ws), ptr(out), c_int(B), c_int(N), c_int(num_heads), _stream(x)), "lvc_mha_mfma_planes")
    return out


def mha_cls(qkv, B, N, num_heads, head_dim, scale):
    """qkv [B*N, 3*H*64] -> [B, H*64]: softmax(q_0 k^T scale) v for the class token of every image (the descriptor network's last block)."""
    _req_cuda(qkv)
    assert head_dim == 64 and N <= 1024
    qkv = qkv.contiguous()
    out = torch.empty(B, num_heads * head_dim, device=qkv.device, dtype=torch.float32)
    check(_lib.lib().lvc_mha_cls(ptr(qkv), ptr(out), c_int(B), c_int(N), c_int(num_heads), c_float(scale), _stream(qkv)), "lvc_mha_cls")
    return out


def mha(qkv, B, N, num_heads, head_dim, scale, mfma=None):
    """qkv [B*N, 3*H*head_dim] -> [B*N, H*head_dim] (softmax(q k^T scale) v per image and head).  Default: the matrix-core
    kernel (lvc_mha_mfma: fp32-accurate two-way fp16 split, probabilities kept in registers between the two products);
    mfma=False, or the range-free split (CONV_SPLIT "bf16x3"): the scalar fp32 kernel.  q * scale, k or v beyond fp16's range
    (or NaN) raises bit 1 of the shared conv error word, like the conv kernels' operands (`check_conv_error_word`)."""
    _req_cuda(qkv)
    qkv = qkv.contiguous()
    out = torch.empty(B * N, num_heads * head_dim, device=qkv.device, dtype=torch.float32)
    if ((MHA_MFMA and CONV_SPLIT == "f16x2") if mfma is None else mfma) and head_dim == 64:
        lib = _lib.lib()
        lib.lvc_mha_workspace_bytes.restype = c_longlong
        ws = torch.empty(max(16, lib.lvc_mha_workspace_bytes(c_int(B), c_int(N), c_int(num_heads))), dtype=torch.uint8, device=qkv.device)
        check(lib.lvc_mha_mfma(ptr(qkv), ptr(out), ptr(ws), c_int(B), c_int(N), c_int(num_heads), c_float(scale),
                                   ptr(_conv_error_view(qkv.device)), _stream(qkv)), "lvc_mha_mfma")
        return out
    check(_lib.lib().lvc_mha(ptr(qkv), ptr(out), c_int(B), c_int(N), c_int(num_heads), c_int(head_dim), c_float(scale), _stream(qkv)),
          "lvc_mha")
    return out
