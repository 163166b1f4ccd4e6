"""Training graph of the action-conditioned SVD UNet on the HIP kernels — SURVEY.md 8(f) row 2 (`FTsvd/train_svd.py:844-970`).

`UNetTrain.forward` evaluates `UNetSpatioTemporalConditionModel.forward` (unet:402-575, micro_cond, one sample per GPU as the
reference's loop) in the UN-FUSED order training needs and records every operator on a tape; `backward` replays the tape in
reverse with the backward building blocks of `train.py` / `csrc/train.hip` and returns fp32 gradients of every live parameter in
the reference's state-dict layout.  The exactly dead parameters of the inference path (single-key cross-attention `norm2`,
`attn2.to_q/to_k`; `add_embedding`) are dead here too and get no gradient — as in the reference's autograd.

Activations are 16-bit (as the forward kernels produce them), parameter gradients fp32; no operator fusion, explicit concat /
residual tensors; the backward kernels (`csrc/train.hip`) are described and measured in DESIGN.md 3.6 / 8.  Tiny host-side pieces
(sinusoidal features, sums of three [T, E] embedding rows, gradient accumulation of sub-64 k-element tensors, weight
re-layouts) are PyTorch plumbing.  Pinned by `tests/golden/train_step_tiny*.npz` (the reference's own `loss.backward()`).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch

from .config import UNetConfig
from .hip import A_CONV3X3, A_CONV3X3_S2, A_CONV_T3, EPI_OUT_F32, Hip
from .train import conv_backward, linear_backward
from .unet import CIN_PAD, sinusoid

SMALL = 1 << 16      # gradients of tensors up to this many elements are accumulated in fp32 on the host side of the tape


class Tape:
    """Operators push a backward closure; `run` seeds the output gradient and replays them in reverse.  Gradients of
    activations are keyed by tensor identity and summed at fan-outs (16-bit `wiw_axpby_bf16`; fp32 for tiny tensors)."""

    def __init__(self, hip: Hip):
        self.hip, self.ops, self.g = hip, [], {}
        self.keep: List[torch.Tensor] = []            # tensors whose id() is a key must stay alive
        self.after_op = None                          # hook: called after every backward closure (gradient hand-over)

    def add(self, t: torch.Tensor, g: torch.Tensor) -> None:
        k = id(t)
        if k not in self.g:
            self.g[k] = g
            self.keep.append(t)
        elif g.numel() <= SMALL or g.dtype == torch.float32:
            self.g[k] = (self.g[k].float() + g.float())
        else:
            self.g[k] = self.hip.axpby(self.g[k], 1.0, g, 1.0)

    def take(self, t: torch.Tensor, dtype=None) -> Optional[torch.Tensor]:
        g = self.g.pop(id(t), None)
        if g is not None and dtype is not None and g.dtype != dtype:
            g = g.to(dtype)
        return g

    def run(self, out: torch.Tensor, dout: torch.Tensor) -> None:
        self.add(out, dout)
        while self.ops:                                # each closure (and the activations it holds) is dropped once it has run
            self.ops.pop()()
            if self.after_op is not None:
                self.after_op()
        self.release()

    def release(self) -> None:
        """Break the tape <-> closure reference cycles NOW: the saved activations of a step (55 GiB at 576x1024x14) must not
        wait for Python's cyclic garbage collector."""
        self.ops.clear()
        self.g.clear()
        self.keep.clear()


def _pad_rows(t: torch.Tensor, mult: int = 64) -> torch.Tensor:
    M = t.shape[0]
    Mp = -(-M // mult) * mult
    if Mp == M:
        return t.contiguous()
    out = t.new_zeros((Mp, *t.shape[1:]))
    out[:M] = t
    return out


class UNetTrain:
    def __init__(self, cfg: UNetConfig, state_dict: Dict[str, torch.Tensor], device="cuda:0", hip: Optional[Hip] = None,
                 dtype: torch.dtype = torch.bfloat16):
        self.cfg = cfg
        self.device = torch.device(device)
        self.hip = hip or Hip(self.device, dtype)
        self.dt = self.hip.dtype
        self.master = {k: torch.as_tensor(v).to(self.device, torch.float32) for k, v in state_dict.items()}
        self.wants = lambda name: True                 # which parameters need a gradient (Trainer: `--train_param_type`)
        self.refresh()

    # ------------------------------------------------------------------------------------------
    # 16-bit operand copies in the kernels' layouts (re-made after every optimiser step)
    # ------------------------------------------------------------------------------------------
    def refresh(self, names=None) -> None:
        """(Re)build the 16-bit GEMM operands from the fp32 masters.  names = None: all of them (construction, checkpoint load,
        the sharded optimiser's all-gather); otherwise only the given parameter names.  `self._direct[name]` is the operand
        (or the slice of a fused operand) that is a plain cast of master[name]: `wiw_adamw_step` writes those itself (`p16`),
        so the single-process step only comes here for the re-laid-out ones (convolutions, padded inputs)."""
        m, dt = self.master, self.dt
        full = names is None
        if full:
            self.W: Dict[str, torch.Tensor] = {}
            self._direct: Dict[str, torch.Tensor] = {}
        for k, v in m.items():
            if not k.endswith(".weight") or v.dim() < 2 or (not full and k not in names):
                continue
            if not full and k in self._direct:                            # a plain cast, possibly into a fused operand's slice
                self._direct[k].copy_(v)
                continue
            if v.dim() == 2:
                w = v
                if w.shape[1] % 64:                                       # add_action_proj.proj: K = 168 -> 192
                    w = torch.cat([w, w.new_zeros(w.shape[0], -w.shape[1] % 64)], dim=1)
                self.W[k] = w.to(dt).contiguous()
                if full and v.shape[1] % 64 == 0:
                    self._direct[k] = self.W[k]
            elif v.dim() == 4:                                            # (O, I, 3, 3) | (O, I, 1, 1) -> [O][ky][kx][I]
                w = v
                if w.shape[1] % 64:                                       # conv_in: 8 -> 64 input channels
                    w = torch.cat([w, w.new_zeros(w.shape[0], -w.shape[1] % 64, *w.shape[2:])], dim=1)
                self.W[k] = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(dt).contiguous()
            else:                                                         # (O, I, 3, 1, 1) -> [O][kt][I]
                self.W[k] = v[:, :, :, 0, 0].permute(0, 2, 1).reshape(v.shape[0], -1).to(dt).contiguous()
        if not full:
            return
        # fused q | k | v projections of the self-attentions: ONE [3C, C] operand, the three reference tensors are its row slices
        for k in list(m):
            if k.endswith(".attn1.to_q.weight"):
                p = k[: -len("to_q.weight")]
                fused = torch.cat([m[p + "to_q.weight"], m[p + "to_k.weight"], m[p + "to_v.weight"]]).to(dt).contiguous()
                self.W[p + "to_qkv.weight"] = fused
                C = m[k].shape[0]
                for i, nm in enumerate(("to_q.weight", "to_k.weight", "to_v.weight")):
                    self.W[p + nm] = self._direct[p + nm] = fused[i * C:(i + 1) * C]

    # ------------------------------------------------------------------------------------------
    # operators (forward + tape entry)
    # ------------------------------------------------------------------------------------------
    def _unit_colsum(self, dy: torch.Tensor, units: int, rows_per_unit: int) -> torch.Tensor:
        return self.hip.colsum(dy, units * rows_per_unit, dy.shape[1], units=units).reshape(units, dy.shape[1])

    def linear(self, x, name, M, bias=True, res=None, rowvec=None, rows_per_vec=1, out_f32=False, wkey=None, split=None):
        """y = x . W^T (+ b) (+ rowvec[row // rows_per_vec]) (+ res).  x [>= M rows, K] 16-bit (extra rows are padding)."""
        hip, tape = self.hip, self.tape
        W = self.W[wkey or name + ".weight"]
        N, K = W.shape
        b = self.master[name + ".bias"] if bias else None
        # 16-bit outputs of a handful of rows are allocated with their rows padded to 64 (zeros): they are the A operand of
        # the next small GEMM and of its weight-gradient GEMM (K loop over rows).  fp32 outputs (per-unit vectors) are exact.
        Mp = 64 if (M < 64 and not out_f32) else M
        y = (torch.zeros if Mp != M else torch.empty)(Mp, N, dtype=torch.float32 if out_f32 else self.dt, device=self.device)
        hip.gemm(x, W, y, M=M, N=N, K=K, C1=K, bias=b, epilogue=EPI_OUT_F32 if out_f32 else 0,
                 res1=res, ldr1=N if res is not None else 0, beta1=1.0 if res is not None else 0.0,
                 rowvec=rowvec, rowvec_ld=N if rowvec is not None else 0, rows_per_vec=rows_per_vec)

        def bwd():
            dy = tape.take(y, self.dt)
            if dy is None:
                return
            if res is not None:
                tape.add(res, dy)
            if rowvec is not None:
                tape.add(rowvec, self._unit_colsum(dy, rowvec.shape[0], rows_per_vec))
            xp, dyp = _pad_rows(x[:M]), _pad_rows(dy[:M])
            need_dw = any(self.wants(nm) for nm in (split or [name + ".weight"]))
            need_db = bias and self.wants(name + ".bias")
            dx, dW, db = linear_backward(hip, xp, W, dyp, need_db=need_db, need_dw=need_dw)
            tape.add(x, dx[:M] if x.shape[0] == M else _pad_like(dx[:M], x))
            if need_dw and split is None:
                self.grads[name + ".weight"] = dW[:, : self.master[name + ".weight"].shape[1]].contiguous()
            elif need_dw:                                                # fused q | k | v: three reference tensors
                for i, nm in enumerate(split):
                    self.grads[nm] = dW[i * (N // 3):(i + 1) * (N // 3)].contiguous()
            if need_db:
                self.grads[name + ".bias"] = db
        tape.ops.append(bwd)
        return y

    def conv(self, x, name, M_out, H, W, mode=A_CONV3X3, T=1, res=None, rowvec=None, rows_per_vec=1, alpha=1.0, wkey=None):
        """3x3 (stride 1 / stride 2) or temporal convolution + bias (+ per-frame vector) (+ residual).  (H, W) = OUTPUT geometry."""
        hip, tape = self.hip, self.tape
        Wk = self.W[wkey or name + ".weight"]
        Cout = Wk.shape[0]
        taps = 3 if mode == A_CONV_T3 else 9
        Cin = Wk.shape[1] // taps
        b = self.master[name + ".bias"]
        y = torch.empty(M_out, Cout, dtype=self.dt, device=self.device)
        hip.gemm(x, Wk, y, M=M_out, N=Cout, K=taps * Cin, C1=Cin, mode=mode, H=H, Wd=W, T=T, bias=b,
                 res1=res, ldr1=Cout if res is not None else 0, beta1=1.0 if res is not None else 0.0,
                 rowvec=rowvec, rowvec_ld=Cout if rowvec is not None else 0, rows_per_vec=rows_per_vec)
        ref_w = self.master[name + ".weight"]

        def bwd():
            dy = tape.take(y, self.dt)
            if dy is None:
                return
            if res is not None:
                tape.add(res, dy)
            if rowvec is not None:
                tape.add(rowvec, self._unit_colsum(dy, rowvec.shape[0], rows_per_vec))
            dyp, Wp = dy, Wk
            if Cout % 64:                                                 # conv_out: 4 output channels -> 64
                dyp = torch.cat([dy, dy.new_zeros(M_out, 64 - Cout)], dim=1).contiguous()
                Wp = torch.cat([Wk, Wk.new_zeros(64 - Cout, Wk.shape[1])]).contiguous()
            need_dw = self.wants(name + ".weight") or self.wants(name + ".bias")
            if mode == A_CONV3X3_S2:
                # dx on the (2H, 2W) grid = stride-1 conv of the dilated dy with mirrored taps; dW from stride-2 im2col rows
                dil = hip.row_map(dyp, hip.ROW_DILATE2X, 4 * M_out, dyp.shape[1], H, W)
                W2 = Wp.reshape(Wp.shape[0], 9, Cin).flip(1).permute(2, 1, 0).reshape(Cin, 9 * Wp.shape[0]).contiguous()
                dx = torch.empty(4 * M_out, Cin, dtype=self.dt, device=self.device)
                hip.gemm(dil, W2, dx, M=4 * M_out, N=Cin, K=9 * Wp.shape[0], C1=Wp.shape[0], mode=A_CONV3X3, H=2 * H, Wd=2 * W)
                if need_dw:
                    from .train import wgrad
                    dW = wgrad(hip, dyp, x, M_out, dyp.shape[1], 9 * Cin, view_ok=True, conv=(Cin, H, W, 1, False, 2))
                    db = hip.colsum(dy, M_out, Cout)
            else:
                dx, dW, db = conv_backward(hip, x, Wp, dyp, H, W, T=T, temporal=(mode == A_CONV_T3), need_dw=need_dw)
                db = db[:Cout] if need_dw else None
            tape.add(x, dx)
            if not need_dw:                                               # frozen convolution: only dx flows on
                return
            dW = dW[:Cout]
            if mode == A_CONV_T3:                                         # unflatten: a view of either orientation -> one copy
                g = dW.unflatten(1, (3, Cin)).permute(0, 2, 1)[:, : ref_w.shape[1]].reshape(ref_w.shape)
            else:
                g = dW.unflatten(1, (3, 3, Cin)).permute(0, 3, 1, 2)[:, : ref_w.shape[1]]
            self.grads[name + ".weight"] = g.contiguous()
            self.grads[name + ".bias"] = db
        tape.ops.append(bwd)
        return y

    def conv1x1(self, x, name, M):
        """conv_shortcut (1x1) of ResnetBlock2D: a linear layer on tokens; gradient reshaped to (O, I, 1, 1)."""
        y = self.linear(x, name, M)
        ref = self.master[name + ".weight"]

        def fix():
            if name + ".weight" in self.grads and self.grads[name + ".weight"].dim() == 2:
                self.grads[name + ".weight"] = self.grads[name + ".weight"].reshape(ref.shape)
        self.tape.ops.insert(len(self.tape.ops) - 1, fix)     # runs AFTER the linear's backward (reverse order)
        return y

    def groupnorm(self, x, name, M, rows_per_unit, eps, silu):
        hip, tape = self.hip, self.tape
        C = x.shape[1]
        g, b = self.master[name + ".weight"], self.master[name + ".bias"]
        y = hip.groupnorm(x, C, None, 0, M, rows_per_unit, g, b, eps, silu)

        def bwd():
            dy = tape.take(y, self.dt)
            if dy is None:
                return
            dx, dg, db = hip.groupnorm_bwd(x, dy, g, b, M, C, rows_per_unit, eps, silu)
            tape.add(x, dx)
            self.grads[name + ".weight"], self.grads[name + ".bias"] = dg, db
        tape.ops.append(bwd)
        return y

    def layernorm(self, x, name, M):
        hip, tape = self.hip, self.tape
        C = x.shape[1]
        g, b = self.master[name + ".weight"], self.master[name + ".bias"]
        y = hip.layernorm(x, M, C, g, b, 1e-5)

        def bwd():
            dy = tape.take(y, self.dt)
            if dy is None:
                return
            dx, dg, db = hip.layernorm_bwd(x, dy, g, M, C, 1e-5)
            tape.add(x, dx)
            self.grads[name + ".weight"], self.grads[name + ".bias"] = dg, db
        tape.ops.append(bwd)
        return y

    def geglu(self, P, M):
        hip, tape = self.hip, self.tape
        Ch = P.shape[1] // 2
        h = hip.geglu_fwd(P, M, Ch)

        def bwd():
            dh = tape.take(h, self.dt)
            if dh is not None:
                tape.add(P, hip.geglu_bwd(P, dh, M, Ch))
        tape.ops.append(bwd)
        return h

    def silu(self, x):
        hip, tape = self.hip, self.tape
        y = hip.silu(x)

        def bwd():
            dy = tape.take(y, self.dt)
            if dy is not None:
                tape.add(x, hip.silu(x, dy))
        tape.ops.append(bwd)
        return y

    def blend(self, xs, xt, name):
        """AlphaBlender (resnet.py:784-797): a xs + (1 - a) xt, a = sigmoid(mix_factor)."""
        hip, tape = self.hip, self.tape
        a = float(torch.sigmoid(self.master[name + ".mix_factor"]).item())
        y = hip.axpby(xs, a, xt, 1.0 - a)

        def bwd():
            dy = tape.take(y, self.dt)
            if dy is None:
                return
            tape.add(xs, hip.axpby(dy, a))
            tape.add(xt, hip.axpby(dy, 1.0 - a))
            self.grads[name + ".mix_factor"] = (hip.dot(dy, xs, xt) * (a * (1.0 - a))).reshape(self.master[name + ".mix_factor"].shape)
        tape.ops.append(bwd)
        return y

    def concat(self, x1, x2):
        tape = self.tape
        y = torch.cat([x1, x2], dim=1).contiguous()                        # a copy (plumbing)
        C1 = x1.shape[1]

        def bwd():
            dy = tape.take(y, self.dt)
            if dy is not None:
                tape.add(x1, dy[:, :C1].contiguous())
                tape.add(x2, dy[:, C1:].contiguous())
        tape.ops.append(bwd)
        return y

    def upsample2x(self, x, M, H, W):
        hip, tape = self.hip, self.tape
        C = x.shape[1]
        y = hip.row_map(x, hip.ROW_UPSAMPLE2X, 4 * M, C, H, W)

        def bwd():
            dy = tape.take(y, self.dt)
            if dy is not None:
                tape.add(x, hip.row_map(dy, hip.ROW_SUMPOOL2X2, M, C, H, W))
        tape.ops.append(bwd)
        return y

    def self_attention(self, a, p, M, seqs, S, heads, temporal_T=0):
        """to_q|k|v (one GEMM) -> softmax(Q K^T / 8) V -> returns O [M, C] (the out-projection is the caller's linear).
        temporal_T > 0: rows are (b, t, s) and the sequences are the T frames of every site: tokens are re-ordered to
        (b, s, t padded to 16) around the same kernels."""
        hip, tape = self.hip, self.tape
        C = heads * 64
        qkv = self.linear(a, p + ".to_qkv", M, bias=False, wkey=p + ".to_qkv.weight",
                          split=(p + ".to_q.weight", p + ".to_k.weight", p + ".to_v.weight"))
        if temporal_T:
            T, Tp, Ssp = temporal_T, 16, M // (seqs * temporal_T)          # seqs = batch items here
            nseq = seqs * Ssp
            q_seq = hip.row_map(qkv, hip.ROW_T_TO_SEQ, nseq * Tp, 3 * C, T=T, Tp=Tp, S=Ssp)
            Ms = nseq * Tp
            vt = torch.empty(C, Ms, dtype=self.dt, device=self.device)
            hip.transpose(q_seq, 3 * C, 2 * C, Ms, C, vt, Ms)
            o_seq = torch.empty(Ms, C, dtype=self.dt, device=self.device)
            hip.attn_small(q_seq, 3 * C, C, vt, Ms, o_seq, C, nseq, T, Tp, heads, 64, 0.125)
            o = hip.row_map(o_seq, hip.ROW_SEQ_TO_T, M, C, T=T, Tp=Tp, S=Ssp)

            def bwd():
                do = tape.take(o, self.dt)
                if do is None:
                    return
                do_seq = hip.row_map(do, hip.ROW_T_TO_SEQ, Ms, C, T=T, Tp=Tp, S=Ssp)
                dq_seq = hip.attn_backward(q_seq, o_seq, do_seq, nseq, T, heads, 0.125, Sp=Tp)
                tape.add(qkv, hip.row_map(dq_seq, hip.ROW_SEQ_TO_T, M, 3 * C, T=T, Tp=Tp, S=Ssp))
            tape.ops.append(bwd)
            return o
        vt = torch.empty(C, M, dtype=self.dt, device=self.device)
        hip.transpose(qkv, 3 * C, 2 * C, M, C, vt, M)
        o = torch.empty(M, C, dtype=self.dt, device=self.device)
        # the forward keeps its row log-sum-exp (4 bytes per row and head) for the backward where the LDS-tiled backward kernels
        # serve the sequence (the spatial levels down to 8 x 16 latents): one Q.K^T pass less per layer in the step
        lse = torch.empty(seqs * heads * S, dtype=torch.float32, device=self.device) if (S % 32 == 0 and S >= 128) else None
        hip.attn_spatial(qkv, 3 * C, C, vt, M, o, C, seqs, S, heads, 0.125, lse=lse)

        def bwd():
            do = tape.take(o, self.dt)
            if do is not None:
                tape.add(qkv, hip.attn_backward(qkv, o, do, seqs, S, heads, 0.125, lse=lse))
        tape.ops.append(bwd)
        return o

    # ------------------------------------------------------------------------------------------
    # small dense paths (a handful of rows): embeddings and the single-key cross-attention vectors
    # ------------------------------------------------------------------------------------------
    def mlp(self, x, p, M, out_f32=True):
        """TimestepEmbedding (embeddings.py:804-816): linear_1 -> SiLU -> linear_2."""
        h = self.linear(x, p + ".linear_1", M)                 # rows padded to 64 by `linear`
        return self.linear(self.silu(h), p + ".linear_2", M, out_f32=out_f32)

    def host_sum(self, terms, shape):
        """fp32 sum of small [T, E] tensors with broadcasting rows (the three embeddings of unet:465-487); plumbing."""
        tape = self.tape
        y = torch.zeros(shape, dtype=torch.float32, device=self.device)
        for t in terms:
            y = y + t.float()

        def bwd():
            dy = tape.take(y)
            if dy is None:
                return
            for t in terms:
                tape.add(t, dy.float().sum(0, keepdim=True) if t.shape[0] == 1 and shape[0] > 1 else dy.float())
        tape.ops.append(bwd)
        return y

    def cast16(self, x, rows_pad=64):
        """fp32 [M, C] -> 16-bit, rows padded to a multiple of 64 (GEMM operand); gradient flows back in fp32."""
        tape = self.tape
        M = x.shape[0]
        y = _pad_rows(x.to(self.dt), rows_pad)

        def bwd():
            dy = tape.take(y)
            if dy is not None:
                tape.add(x, dy[:M].float())
        tape.ops.append(bwd)
        return y

    # ------------------------------------------------------------------------------------------
    # blocks
    # ------------------------------------------------------------------------------------------
    def res_block(self, p, x, Cin, Cout, M, H, W, T, emb_silu, eps):
        """SpatioTemporalResBlock (resnet.py:686-716)."""
        S = H * W
        s, t = p + ".spatial_res_block", p + ".temporal_res_block"
        te_s = self.linear(emb_silu, s + ".time_emb_proj", T, out_f32=True)          # [T, Cout] per-frame vectors
        h = self.groupnorm(x, s + ".norm1", M, S, eps, True)
        h = self.conv(h, s + ".conv1", M, H, W, rowvec=te_s, rows_per_vec=S)
        h = self.groupnorm(h, s + ".norm2", M, S, eps, True)
        sc = self.conv1x1(x, s + ".conv_shortcut", M) if (s + ".conv_shortcut.weight") in self.master else x
        xs = self.conv(h, s + ".conv2", M, H, W, res=sc)
        te_t = self.linear(emb_silu, t + ".time_emb_proj", T, out_f32=True)
        h = self.groupnorm(xs, t + ".norm1", M, T * S, eps, True)
        h = self.conv(h, t + ".conv1", M, H, W, mode=A_CONV_T3, T=T, rowvec=te_t, rows_per_vec=S)
        h = self.groupnorm(h, t + ".norm2", M, T * S, eps, True)
        xt = self.conv(h, t + ".conv2", M, H, W, mode=A_CONV_T3, T=T, res=xs)
        return self.blend(xs, xt, p + ".time_mixer")

    def cross_vector(self, q, ehs16):
        """Single-key cross-attention: softmax over one key == 1, so attn2(x) = to_out(to_v(ctx)), one [1, C]
        vector; norm2 / to_q / to_k receive no gradient."""
        v = self.linear(ehs16, q + ".attn2.to_v", 1, bias=False)
        return self.linear(v, q + ".attn2.to_out.0", 1, out_f32=True)

    def ff(self, a, p, M, **epi):
        P = self.linear(a, p + ".net.0.proj", M)
        return self.linear(self.geglu(P, M), p + ".net.2", M, **epi)

    def transformer(self, p, x, C, M, H, W, T, heads, ehs16):
        """TransformerSpatioTemporalModel (transformer_temporal.py:279-382), one layer, one sample."""
        S = H * W
        b, t = p + ".transformer_blocks.0", p + ".temporal_transformer_blocks.0"
        xn = self.groupnorm(x, p + ".norm", M, S, 1e-6, False)
        h = self.linear(xn, p + ".proj_in", M)
        # spatial block (attention.py:462-582)
        o = self.self_attention(self.layernorm(h, b + ".norm1", M), b + ".attn1", M, T, S, heads)
        h = self.linear(o, b + ".attn1.to_out.0", M, res=h, rowvec=self.cross_vector(b, ehs16), rows_per_vec=M)
        feat = torch.from_numpy(sinusoid(np.arange(T), C)).to(self.device, self.dt)
        pos = self.mlp(_pad_rows(feat), p + ".time_pos_embed", T)                 # [T, C] frame-position embedding
        hs = self.ff(self.layernorm(h, b + ".norm3", M), b + ".ff", M, res=h)
        # temporal block on hs + emb (attention.py:707-762); rows stay (t, s): every operator but the attention is per token
        hm0 = self._add_rowvec(hs, pos, S)
        hm = self.ff(self.layernorm(hm0, t + ".norm_in", M), t + ".ff_in", M, res=hm0)
        o = self.self_attention(self.layernorm(hm, t + ".norm1", M), t + ".attn1", M, 1, S, heads, temporal_T=T)
        hm = self.linear(o, t + ".attn1.to_out.0", M, res=hm, rowvec=self.cross_vector(t, ehs16), rows_per_vec=M)
        ht = self.ff(self.layernorm(hm, t + ".norm3", M), t + ".ff", M, res=hm)
        hb = self.blend(hs, ht, p + ".time_mixer")
        return self.linear(hb, p + ".proj_out", M, res=x)

    def _add_rowvec(self, x, vec, rows_per_vec):
        """x + vec[row // rows_per_vec] (vec fp32 [units, C]): a 1x... broadcast add through the GEMM-free path."""
        hip, tape = self.hip, self.tape
        rep = vec.to(self.dt).repeat_interleave(rows_per_vec, dim=0).contiguous()      # plumbing: broadcast copy
        y = hip.axpby(x, 1.0, rep, 1.0)

        def bwd():
            dy = tape.take(y, self.dt)
            if dy is not None:
                tape.add(x, dy)
                tape.add(vec, self._unit_colsum(dy, vec.shape[0], rows_per_vec))
        tape.ops.append(bwd)
        return y

    # ------------------------------------------------------------------------------------------
    # the network
    # ------------------------------------------------------------------------------------------
    def forward(self, sample: torch.Tensor, timestep: float, ehs: torch.Tensor, added_time_ids: torch.Tensor,
                action_ids: torch.Tensor) -> torch.Tensor:
        """sample (1,T,8,h,w); ehs (1,1,Dctx); added_time_ids (1,3); action_ids (1,T,Ch).  Returns (1,T,4,h,w) fp32."""
        cfg, hip, dev, dt = self.cfg, self.hip, self.device, self.dt
        if getattr(self, "tape", None) is not None:
            self.tape.release()                         # a forward without a backward (evaluation) leaves its tape behind
        self.tape, self.grads = Tape(hip), {}
        T, ch, n, L = cfg.num_frames, cfg.block_out_channels, len(cfg.block_out_channels), cfg.layers_per_block
        _, _, cin, h, w = sample.shape
        assert sample.shape[0] == 1 and sample.shape[1] == T
        E = cfg.time_embed_dim
        # ---- conditioning embeddings (unet:447-487), a handful of rows
        t_feat = torch.from_numpy(sinusoid(np.array([timestep], np.float32), ch[0])).to(dev, dt)
        emb_t = self.mlp(_pad_rows(t_feat), "time_embedding", 1)
        a = action_ids.float().reshape(T, -1)
        feats = []
        for k in (1.0, 2.0, 4.0, 6.0, 8.0, 10.0):
            feats += [torch.cos(k * a), torch.sin(k * a)]
        act = torch.stack(feats, dim=-1).reshape(T, -1)
        Kp = self.W["add_action_proj.proj.weight"].shape[1]
        act = torch.cat([act, act.new_zeros(T, Kp - act.shape[1])], dim=1).to(dev, dt)
        emb_a = self.mlp(self.linear(_pad_rows(act), "add_action_proj.proj", T), "add_embedding_action", T)
        n_feat = torch.from_numpy(sinusoid(added_time_ids[:, -1].float().numpy(), cfg.addition_time_embed_dim)).to(dev, dt)
        emb_n = self.mlp(_pad_rows(n_feat), "add_embedding_noise", 1)
        emb = self.host_sum([emb_t, emb_a, emb_n], (T, E))
        emb_silu = self.silu(self.cast16(emb))                                     # [64, E], rows >= T are silu(0) = 0
        ehs16 = _pad_rows(ehs.reshape(1, -1).to(dev, dt))
        # ---- conv_in
        x0 = torch.zeros(T * h * w, CIN_PAD, dtype=dt, device=dev)
        x0[:, :cin] = sample[0].permute(0, 2, 3, 1).reshape(-1, cin).to(dev, dt)
        H, W = h, w
        M = T * H * W
        x = self.conv(x0, "conv_in", M, H, W)
        skips = [(x, ch[0])]
        C = ch[0]
        for i in range(n):
            p = f"down_blocks.{i}"
            has_attn = i < n - 1
            eps = 1e-6 if has_attn else 1e-5
            for j in range(L):
                x = self.res_block(f"{p}.resnets.{j}", x, C, ch[i], M, H, W, T, emb_silu, eps)
                C = ch[i]
                if has_attn:
                    x = self.transformer(f"{p}.attentions.{j}", x, C, M, H, W, T, cfg.num_attention_heads[i], ehs16)
                skips.append((x, C))
            if i < n - 1:
                H, W = H // 2, W // 2
                M = T * H * W
                x = self.conv(x, f"{p}.downsamplers.0.conv", M, H, W, mode=A_CONV3X3_S2)
                skips.append((x, C))
        x = self.res_block("mid_block.resnets.0", x, C, C, M, H, W, T, emb_silu, 1e-5)
        x = self.transformer("mid_block.attentions.0", x, C, M, H, W, T, cfg.num_attention_heads[-1], ehs16)
        x = self.res_block("mid_block.resnets.1", x, C, C, M, H, W, T, emb_silu, 1e-5)
        rch = list(reversed(ch))
        rheads = list(reversed(cfg.num_attention_heads))
        for i in range(n):
            p = f"up_blocks.{i}"
            for j in range(L + 1):
                sk, Cs = skips.pop()
                x = self.res_block(f"{p}.resnets.{j}", self.concat(x, sk), C + Cs, rch[i], M, H, W, T, emb_silu, 1e-6)
                C = rch[i]
                if i > 0:
                    x = self.transformer(f"{p}.attentions.{j}", x, C, M, H, W, T, rheads[i], ehs16)
            if i < n - 1:
                x = self.upsample2x(x, M, H, W)
                H, W = 2 * H, 2 * W
                M = T * H * W
                x = self.conv(x, f"{p}.upsamplers.0.conv", M, H, W)
        x = self.groupnorm(x, "conv_norm_out", M, H * W, 1e-5, True)
        y = self.conv(x, "conv_out", M, H, W)                                       # [M, 4] 16-bit
        self._out = y
        return y.float().reshape(T, H, W, -1).permute(0, 3, 1, 2).unsqueeze(0)

    def backward(self, dpred: torch.Tensor, loss_scale: float = 1.0) -> Dict[str, torch.Tensor]:
        """dpred: dL/d(model_pred) (1,T,4,h,w) fp32 -> {reference parameter name: fp32 gradient}.
        loss_scale: the 16-bit activation gradients are computed for loss_scale * L and the fp32 parameter gradients divided
        by it at the end — fp16's 6e-5 normal range loses the ~1e-5 activation gradients of a mean loss without it (the
        reference runs fp16 under accelerate's GradScaler, train_svd.py:699, 971-975); bf16 needs none."""
        dy = (dpred[0].permute(0, 2, 3, 1).reshape(-1, dpred.shape[2]) * loss_scale).to(self.device, self.dt).contiguous()
        self.tape.run(self._out, dy)
        self._out = None
        if loss_scale != 1.0:
            inv = 1.0 / loss_scale
            self.grads = {k: v.float() * inv for k, v in self.grads.items()}
        return self.grads


class EMAShadow:
    """`diffusers.training_utils.EMAModel` as train_svd.py uses it (`--use_ema`, :566-568: every argument at its default, :979-980
    one `step` per synchronised optimiser step): fp32 shadow copies of ALL parameters, updated by
        shadow -= (1 - decay_t) * (shadow - param)          for parameters that train (`requires_grad`),
        shadow  = param                                     for frozen ones (`--train_param_type`),
    decay_t = min((1 + s) / (10 + s), decay) with s = optimization_step - update_after_step - 1 (0 while s <= 0; the warm-up
    form 1 - (1 + s / inv_gamma) ** -power under `use_ema_warmup`), clamped below by min_decay (training_utils.py:405-422).
    The update runs on `wiw_ema_step_f32` (the reference's three fp32 operations, bit for bit) when a `Hip` is given, else on
    the same torch expression (CPU tests).  Every rank keeps the full shadow, as under the reference's ZeRO-1."""

    def __init__(self, params: Dict[str, torch.Tensor], trainable=None, hip=None, decay: float = 0.9999, min_decay: float = 0.0,
                 update_after_step: int = 0, use_ema_warmup: bool = False, inv_gamma: float = 1.0, power: float = 2 / 3):
        self.shadow = {k: v.detach().to(torch.float32).contiguous().clone() for k, v in params.items()}   # flat views below
        self.trainable = trainable or (lambda n: True)
        self.hip = hip
        self.decay, self.min_decay, self.update_after_step = float(decay), float(min_decay), int(update_after_step)
        self.use_ema_warmup, self.inv_gamma, self.power = bool(use_ema_warmup), inv_gamma, power
        self.optimization_step = 0
        self.cur_decay_value = None
        self._stored = None

    def get_decay(self, optimization_step: int) -> float:
        step = max(0, optimization_step - self.update_after_step - 1)
        if step <= 0:
            return 0.0
        if self.use_ema_warmup:
            cur = 1 - (1 + step / self.inv_gamma) ** -self.power
        else:
            cur = (1 + step) / (10 + step)
        return max(min(cur, self.decay), self.min_decay)

    @torch.no_grad()
    def step(self, params: Dict[str, torch.Tensor]) -> None:
        self.optimization_step += 1
        decay = self.get_decay(self.optimization_step)
        self.cur_decay_value = decay
        omd = 1 - decay
        for k, s in self.shadow.items():
            p = params[k]
            if not self.trainable(k):
                s.copy_(p)
            elif self.hip is not None and s.is_cuda:
                self.hip.ema_step(s.view(-1), p.reshape(-1), omd)
            else:
                s.sub_(omd * (s - p))

    def state(self) -> dict:
        """The non-tensor part of `EMAModel.state_dict()` — what `save_pretrained` registers into unet_ema/config.json."""
        return {"decay": self.decay, "min_decay": self.min_decay, "optimization_step": self.optimization_step,
                "update_after_step": self.update_after_step, "use_ema_warmup": self.use_ema_warmup, "inv_gamma": self.inv_gamma,
                "power": self.power}

    def load(self, shadow: Dict[str, torch.Tensor], state: dict) -> None:
        assert set(shadow) == set(self.shadow), "unet_ema parameters do not match this architecture"
        for k, v in shadow.items():
            self.shadow[k].copy_(v)
        from .checkpoint import EMA_STATE_KEYS

        for k in EMA_STATE_KEYS:        # the EMA state only: nothing else in a config.json may overwrite an attribute
            if k in state:
                setattr(self, k, state[k])

    # validation under the averaged weights (train_svd.py:1004-1007, 1189-1191): store -> copy_to -> ... -> restore
    @torch.no_grad()
    def store(self, params: Dict[str, torch.Tensor]) -> None:
        self._stored = {k: v.detach().clone() for k, v in params.items()}

    @torch.no_grad()
    def copy_to(self, params: Dict[str, torch.Tensor]) -> None:
        for k, s in self.shadow.items():
            params[k].copy_(s)

    @torch.no_grad()
    def restore(self, params: Dict[str, torch.Tensor]) -> None:
        assert self._stored is not None, "restore() without store()"
        for k, v in self._stored.items():
            params[k].copy_(v)
        self._stored = None


class Trainer:
    """One fine-tuning step of the reference loop (`FTsvd/train_svd.py:844-970`) on one GPU, one sample per step:
        prepare_step -> UNetTrain.forward -> wiw_edm_loss_grad -> UNetTrain.backward -> AdamW -> refreshed 16-bit operands.
    Single process: `wiw_adamw_step` over every parameter tensor.  Data parallel: hand `optimizer=parallel.ShardedAdamW(...)`
    (gradients are copied into its flat buffer, reduced-scattered, the owned slices updated, parameters all-gathered)."""

    PARAM_TYPES = {"full": lambda n: True,
                   "new": lambda n: ("action" in n) or ("noise" in n),
                   "new+temp_layer": lambda n: ("temporal_transformer_block" in n) or ("action" in n) or ("noise" in n)}

    @staticmethod
    def is_dead(name: str) -> bool:
        """Parameters that never receive a gradient under `micro_cond` (in the reference's autograd too: SURVEY.md 9.3): the
        query / key side and the LayerNorm of the single-key cross-attentions, and `add_embedding` (overwritten, unet:458-482)."""
        return ("attn2.to_q." in name or "attn2.to_k." in name or name.startswith("add_embedding.")
                or ("transformer_blocks." in name and ".norm2." in name))

    @classmethod
    def optimizer_shapes(cls, net: "UNetTrain", train_param_type: str = "full") -> Dict[str, tuple]:
        """Shapes (in model order) of the parameters an optimiser has to hold: trainable under `--train_param_type` AND live.
        Build `parallel.ShardedAdamW` from this: torch.optim.AdamW leaves parameters without a gradient untouched, a flat
        optimiser over ALL parameters would apply its decoupled weight decay to the frozen / dead ones on every step (and
        spend master / moment memory and collective bandwidth on 1.5 B parameters when only a few million train)."""
        want = cls.PARAM_TYPES[train_param_type]
        return {k: tuple(v.shape) for k, v in net.master.items() if want(k) and not cls.is_dead(k)}

    def __init__(self, net: "UNetTrain", lr: float = 1e-5, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 optimizer=None, loss_scale: Optional[float] = None, train_param_type: str = "full", grad_accum: int = 1,
                 autotune: bool = False, scale_growth_interval: int = 2000, use_ema: bool = False, ema_kwargs: Optional[dict] = None):
        self.net, self.lr, self.betas, self.eps, self.wd = net, lr, betas, eps, weight_decay
        # `--gradient_accumulation_steps` (train_svd.sh:20 runs 4; accelerate averages the micro-batch losses, train_svd.py:
        # 864, 961-969): `step` is one micro-batch, the optimiser runs on every grad_accum-th call with the mean gradient
        self.grad_accum, self._micro, self._acc = int(grad_accum), 0, {}
        if autotune:                                   # measured plans for the weight-gradient GEMMs (train.wgrad_gemm)
            from .train import set_wgrad_tuning
            set_wgrad_tuning(True)
        # which parameters are updated — the reference's `--train_param_type` (train_svd.py:655-663)
        base = self.PARAM_TYPES[train_param_type]
        owned = None if optimizer is None else set(optimizer.offsets)          # a sharded optimiser may hold a subset
        self.trainable = base if owned is None else (lambda n: base(n) and n in owned)
        net.wants = self.trainable                     # frozen weights: their weight-gradient GEMMs are not launched
        # dynamic loss scale for fp16, as torch.cuda.amp.GradScaler (train_svd.py:960-964 through accelerate): halved — and the
        # step skipped — when a gradient comes back non-finite, doubled after `scale_growth_interval` consecutive good
        # optimiser steps (GradScaler's growth_interval = 2000), capped at 2^16; 1 for bf16 (no scaling)
        self.loss_scale = (2.0 ** 14 if net.dt == torch.float16 else 1.0) if loss_scale is None else loss_scale
        # scaling is a property of the RUN (fp16 storage, or an explicit scale), not of the scale's current value: an fp16 run
        # that has backed off to 1.0 keeps checking for overflow and grows again.  (GradScaler starts at 2^16 and has no cap;
        # here 2^14 and a cap of 2^16: the gradients of this network overflow fp16 above that, so the first steps are not
        # spent backing off.)
        self.scaling = net.dt == torch.float16 or self.loss_scale != 1.0
        self.scale_growth_interval, self._good_steps = int(scale_growth_interval), 0
        self.opt = optimizer
        self.steps = 0
        self.m = {k: torch.zeros_like(v) for k, v in net.master.items()} if optimizer is None else None
        self.v = {k: torch.zeros_like(v) for k, v in net.master.items()} if optimizer is None else None
        self._seen = None                                      # names that got a gradient in the previous step
        if optimizer is not None:
            optimizer.load(net.master)
        # `--use_ema` (train_svd.py:566-568): requires_grad is the `--train_param_type` predicate, not the optimiser's subset
        self.ema = EMAShadow(net.master, trainable=base, hip=net.hip, **(ema_kwargs or {})) if use_ema else None

    def save(self, output_dir: str, total_limit: Optional[int] = None) -> str:
        """`accelerator.save_state(checkpoint-<global_step>)` (train_svd.py:1032-1062): the fp32 parameters in the reference's
        `unet/` layout, the AdamW moments (this rank's ZeRO-1 slices when sharded), the counters.  See `checkpoint.py`."""
        from . import checkpoint as C

        from .train import wgrad_plans

        meta = {"micro": self._micro, "loss_scale": self.loss_scale, "good_steps": self._good_steps, "grad_accum": self.grad_accum,
                "lr": self.lr, "wgrad_plans": {k: list(v) for k, v in wgrad_plans().items()}}
        ucfg = C.unet_config_dict(self.net.cfg) if hasattr(self.net, "cfg") else None     # unet/config.json, as save_pretrained
        ema = None if self.ema is None else {"shadow": self.ema.shadow, "state": self.ema.state()}
        if self.opt is None:
            optim = {f"exp_avg.{k}": v for k, v in self.m.items()}
            optim.update({f"exp_avg_sq.{k}": v for k, v in self.v.items()})
            return C.save_checkpoint(output_dir, self.steps, self.net.master, optim, dict(meta, world=1), total_limit=total_limit,
                                     unet_config=ucfg, ema=ema)
        o = self.opt
        optim = {"master": o.master, "exp_avg": o.m, "exp_avg_sq": o.v}
        return C.save_checkpoint(output_dir, self.steps, self.net.master if o.rank == 0 else None, optim,
                                 dict(meta, world=o.world, opt_layout=o.layout_fingerprint()), rank=o.rank, sharded=True,
                                 total_limit=total_limit, unet_config=ucfg, ema=ema)

    def load(self, path: str) -> None:
        """Resume from a directory written by `save` (`checkpoint.resolve_resume` maps "latest" to one)."""
        from . import checkpoint as C

        sharded = self.opt is not None
        master, optim, meta = C.load_checkpoint(path, rank=self.opt.rank if sharded else 0, sharded=sharded)
        assert meta["world"] == (self.opt.world if sharded else 1), "the optimizer state was saved for another world size"
        if sharded:   # the per-rank slices only mean something under the SAME flat layout (bucket size, parameter order and set)
            if "opt_layout" not in meta:   # written before the fingerprint existed: the slice lengths are all that can be checked
                import warnings

                warnings.warn("checkpoint carries no optimizer layout fingerprint (older writer): checking slice lengths only")
                assert optim["master"].numel() == self.opt.master.numel(), "optimizer slice length differs from this layout"
            else:
                assert meta["opt_layout"] == self.opt.layout_fingerprint(), \
                    "the sharded optimizer state was saved under another flat layout (bucket size / parameter set or order)"
        assert set(master) == set(self.net.master), "checkpoint parameters do not match this architecture"
        for k, v in master.items():
            self.net.master[k].copy_(v)
        if sharded:
            self.opt.load(self.net.master)
            self.opt.master.copy_(optim["master"]); self.opt.m.copy_(optim["exp_avg"]); self.opt.v.copy_(optim["exp_avg_sq"])
            self.opt.steps = int(meta["global_step"])
        else:
            for k in self.m:
                self.m[k].copy_(optim[f"exp_avg.{k}"]); self.v[k].copy_(optim[f"exp_avg_sq.{k}"])
        from .train import load_wgrad_plans

        load_wgrad_plans(meta.get("wgrad_plans", {}))          # the resumed run keeps the saved run's summation orders
        self.steps, self._micro, self.loss_scale = int(meta["global_step"]), int(meta["micro"]), float(meta["loss_scale"])
        self._good_steps = int(meta.get("good_steps", 0))
        self._acc, self._seen = {}, None
        if self.ema is not None:                              # train_svd.py:600-604: the averaged weights resume from unet_ema/
            self.ema.load(*C.load_ema(path))
        self.net.refresh()

    def fit(self, batches, max_train_steps: int, validation_steps: int = 0, val_samples=None, checkpointing_steps: int = 0,
            output_dir: Optional[str] = None, checkpoints_total_limit: Optional[int] = None, lr_schedule=None, log_path=None,
            val_kwargs: Optional[dict] = None):
        """The outer loop of train_svd.py:844-1062 around `step`: micro-batches from `batches` until `max_train_steps` optimiser
        steps; after every optimiser step (`accelerator.sync_gradients`, :971) the reference's order — checkpoint every
        `checkpointing_steps` (:986-993), validation every `validation_steps` AND at step 1 (:995-1030).  lr_schedule:
        global_step -> lr (`train.lr_at`).  Returns the list of {"step", "loss" | validation dict} records (also appended to
        `log_path` as JSON lines: what `accelerator.log` receives)."""
        import json

        log = []

        def emit(rec):
            log.append(rec)
            if log_path:
                with open(log_path, "a") as f:
                    f.write(json.dumps(rec) + "\n")

        for st in batches:
            if self.steps >= max_train_steps:
                break
            if lr_schedule is not None:
                self.lr = float(lr_schedule(self.steps))
            before = self.steps
            loss = self.step(st)
            if self.steps == before:          # accumulating micro-batch, or a skipped (overflowed) window: no optimiser step
                continue
            emit({"step": self.steps, "train_loss": loss, "lr": self.lr})
            if checkpointing_steps and output_dir and self.steps % checkpointing_steps == 0:
                self.save(output_dir, checkpoints_total_limit)
            if validation_steps and val_samples and (self.steps % validation_steps == 0 or self.steps == 1):
                emit(self.validate(val_samples, **(val_kwargs or {})))
        return log

    @torch.no_grad()
    def validate(self, samples, num_steps: int = 25, use_ema: Optional[bool] = None, frontend=None, log_path: Optional[str] = None,
                 residual_fp32: bool = False) -> dict:
        """The validation pass of the training loop (train_svd.py:996-1030 -> eval_inference :1140-1193): the INFERENCE loop on
        the current weights — the EMA weights under `--use_ema` (:1004-1007; the live ones come back afterwards, :1189-1191) —
        over `samples`, with the reference's knobs (fps 7, motion bucket 127, noise_aug 0.02), and its per-clip PSNR
        (evaluation/FVD/calculate_psnr.py:6-15, `only_final`: mean over clips and frames).  FVD / LPIPS need the I3D / AlexNet
        checkpoints the tree does not hold: not computed.
        samples: dicts with image_latents (1,4,h,w), image_embeddings (1,1,D), noise (1,T,4,h,w), actions (1,T) and the
        ground truth `target_latents` (1,T,4,h,w) [+ `target_frames` (1,T,3,H,W) in [0,1] when a `frontend` decodes].
        Returns {"global_step", "latent_mse", "latent_psnr", ["psnr"], "clips"}; appended as one JSON line to `log_path`
        (`accelerator.log(log_dict, step=global_step)`)."""
        import json
        import math

        from .pipeline import SVDDenoiser
        from .unet import UNetHIP

        use_ema = (self.ema is not None) if use_ema is None else use_ema
        assert not use_ema or self.ema is not None, "validate(use_ema=True) needs Trainer(use_ema=True)"
        weights = self.ema.shadow if use_ema else self.net.master
        unet = UNetHIP(self.net.cfg, weights, self.net.device, hip=self.net.hip, residual_fp32=residual_fp32)
        den = SVDDenoiser(unet, use_graph=False)
        se, n, psnrs, lat_psnrs = 0.0, 0, [], []
        for s in samples:
            lat = den.denoise(s["image_latents"], s["image_embeddings"], s["noise"], s["actions"], num_steps=num_steps, fps=7,
                              motion_bucket_id=127, noise_aug_strength=0.02)
            tgt = torch.as_tensor(s["target_latents"]).to(lat.device, torch.float32)
            d = (lat - tgt).double()
            se += float(d.pow(2).sum()); n += d.numel()
            rng = float(tgt.max() - tgt.min()) or 1.0
            mse = float(d.pow(2).mean())
            lat_psnrs.append(100.0 if mse < 1e-10 else 20 * math.log10(rng / math.sqrt(mse)))
            if frontend is not None and "target_frames" in s:
                fr = torch.as_tensor(frontend.decode(lat.cpu().numpy() if not hasattr(frontend, "decode_frames") else lat))
                fr = (fr.float().cpu() / 2 + 0.5).clamp(0, 1)                    # postprocess_video (video_processor.py:90-113)
                gt = torch.as_tensor(s["target_frames"]).float().cpu()
                for a, b in zip(fr.reshape(-1, *fr.shape[-3:]), gt.reshape(-1, *gt.shape[-3:])):
                    m = float((a - b).pow(2).mean())
                    psnrs.append(100.0 if m < 1e-10 else 20 * math.log10(1.0 / math.sqrt(m)))
        out = {"global_step": self.steps, "clips": len(lat_psnrs), "weights": "ema" if use_ema else "live",
               "latent_mse": se / max(n, 1), "latent_psnr": float(sum(lat_psnrs) / max(len(lat_psnrs), 1))}
        if psnrs:
            out["psnr"] = float(sum(psnrs) / len(psnrs))
        if log_path:
            with open(log_path, "a") as f:
                f.write(json.dumps(out) + "\n")
        del den, unet
        torch.cuda.empty_cache()
        return out

    def _mean_grad(self, name: str, g: torch.Tensor) -> torch.Tensor:
        """This micro-batch's share of the mean gradient plus what the earlier micro-batches of the window left."""
        if self.grad_accum == 1:
            return g
        g = g.reshape(self.net.master[name].shape) * (1.0 / self.grad_accum)
        acc = self._acc.get(name)
        return g if acc is None else acc + g

    def step(self, st) -> float:
        """st: `train.StepInputs` of one micro-batch.  Returns its loss (host float)."""
        from .train import TrainStep

        net, hip = self.net, self.net.hip
        pred = net.forward(st.unet_input, st.timestep, st.ehs, st.added_time_ids, st.action_ids)
        loss, dpred = TrainStep(hip).loss_and_grad(pred, st)
        if self.opt is not None and not self.scaling and self._seen is not None:
            # hand every finished gradient to the sharded optimiser DURING the backward: its buckets are reduce-scattered
            # (asynchronously, over all xGMI links) while the remaining operators still run.  Which parameters get a
            # gradient is learnt from the first step (dead / frozen ones never do and must not be waited for).
            sent = set()
            expected = self._seen

            def hand_over():
                for name in list(net.grads):
                    if name not in sent and self.trainable(name):
                        sent.add(name)
                        self.opt.notify(name, self._mean_grad(name, net.grads[name]), expected)
            if (self._micro + 1) % self.grad_accum == 0:                   # earlier micro-batches accumulate locally (no_sync)
                net.tape.after_op = hand_over
        overlapped = net.tape.after_op is not None
        grads = net.backward(dpred.reshape(pred.shape), self.loss_scale)
        net.tape.after_op = None
        self._seen = {n for n in grads if self.trainable(n)}
        if self.scaling:
            bad = ~torch.stack([torch.isfinite(g).all() for g in grads.values()]).all()
            if self.opt is not None and self.opt.world > 1:                # EVERY rank skips, or none: the collectives of
                import torch.distributed as dist                           # `opt.step()` must be entered by all of them

                flag = bad.to(torch.float32).reshape(1)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                bad = flag[0] > 0
            if bool(bad):                                                  # (one host sync)
                # overflow: skip the update and halve the scale, as a GradScaler does.  The micro-batches already accumulated
                # in this window go with it: under accelerate the non-finite values are IN the accumulated .grad, the
                # optimiser step of the window is skipped and the gradients are zeroed (train_svd.py:961-969)
                self.loss_scale *= 0.5
                self._good_steps = 0
                self._micro, self._acc = 0, {}
                if self.ema is not None:                                   # the window is over (`accelerator.sync_gradients`): the
                    self.ema.step(net.master)                              # reference steps the EMA on a skipped update too (:979)
                return float(loss)
        self._micro += 1
        if self._micro % self.grad_accum != 0:                             # not the last micro-batch: accumulate, no update
            inv = 1.0 / self.grad_accum
            for name, g in grads.items():
                if self.trainable(name):
                    g = g.reshape(net.master[name].shape) * inv
                    self._acc[name] = g if name not in self._acc else self._acc[name] + g
            net.grads = {}
            return float(loss)
        if not overlapped and self.grad_accum > 1:
            grads = {name: self._mean_grad(name, g) for name, g in grads.items() if self.trainable(name)}
        self._acc = {}
        self.steps += 1
        if self.scaling:                                                   # GradScaler growth: x2 after N good steps in a row
            self._good_steps += 1
            if self._good_steps >= self.scale_growth_interval:
                self.loss_scale, self._good_steps = min(self.loss_scale * 2.0, 2.0 ** 16), 0
        if self.opt is None:
            stale = set()
            for name, g in grads.items():                                  # parameters without a gradient (the dead ones) stay
                if not self.trainable(name):
                    continue
                p, p16 = net.master[name], net._direct.get(name)            # plain-cast operands are refreshed by the kernel itself
                hip.adamw_step(p.view(-1), g.reshape(-1).contiguous(), self.m[name].view(-1), self.v[name].view(-1), self.steps,
                               self.lr, self.betas[0], self.betas[1], self.eps, self.wd, p16=None if p16 is None else p16.view(-1))
                if p16 is None:
                    stale.add(name)
            net.refresh(stale)                                             # the re-laid-out operands (convolutions, padded inputs)
            if self.ema is not None:
                self.ema.step(net.master)
            return float(loss)
        else:
            if not overlapped:                                           # first step, or fp16 (un-scaled after the backward)
                for name, g in grads.items():
                    if self.trainable(name):
                        self.opt.view(self.opt.grads, name).copy_(g.reshape(net.master[name].shape))
            self.opt.lr = self.lr                                       # the caller's schedule (`train.lr_at`) sets Trainer.lr
            self.opt.step()
            for name in grads:                                           # parameters without a gradient / frozen ones are not
                if self.trainable(name):                                 # read back (torch.optim.AdamW skips them too)
                    net.master[name].copy_(self.opt.view(self.opt.params, name))
        net.refresh()
        if self.ema is not None:
            self.ema.step(net.master)
        return float(loss)


def _pad_like(t: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    out = torch.zeros_like(like, dtype=t.dtype)
    out[: t.shape[0]] = t
    return out
