"""Batched greedy decoding with a KV cache for WhisperDecoder on MI355X.

New capability (the reference decodes nothing for Whisper: README.md:86).  Semantics = the reference's
generic greedy loop (text/generator.py:23-35): at each step take argmax of the last position's logits and
append it; here for a whole batch, with the self-attention K/V cached and the cross-attention K/V projected
once.  The stop rule is build-defined (no tokenizer in the reference): a fixed number of new tokens.

One step is ~8 launches per layer (csrc/decode.hip); the position and current tokens live on the device, so the
step is captured ONCE into a HIP graph and replayed - no tracing compiler, no host round trips in the loop.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch
from torch import Tensor

from .._hip import DecLayer, check, lib, ops
from ..transformer import _f32


# path="auto" = the launch-per-stage list.  Measured (tools/decode_paths_bench.py, MI355X, batch 32, us per step): Whisper-base 461
# launches / 623 persistent, small 1044 / 1665, large-v2 4476 / 7160 - inside one launch a stage boundary costs MORE than a
# kernel boundary (poll + barrier + sc1 round trips ~4-5 us against ~1.5 us + first loads), exactly what
# cdna_hip_programming.md 5.6 reports for latency-bound phases.  The persistent path stays selectable and tested.
PERSISTENT_BY_DEFAULT = False


def _ptr(t: Tensor | None):
    return None if t is None else t.data_ptr()


@dataclass
class WhisperRules:
    """Whisper's decoding-time logit filters (pm_dec_whisper_rules): ids of the tokenizer in use.  ``blank`` = ids forbidden as
    the first generated token (the blank token and end-of-text); ``max_initial_timestamp`` counts timestamp steps (0.02 s each)
    from ``timestamp_begin``, < 0 = no cap."""
    eot: int
    timestamp_begin: int
    no_timestamps: int = -1
    max_initial_timestamp: int = -1
    suppress: tuple = ()
    blank: tuple = ()


class GreedyDecoder:
    """State + launch list of the decode step for one (decoder, batch, memory length) geometry."""

    def __init__(self, dec, memory: Tensor, prompt: Tensor, n_new: int, margins: bool = False, fused: bool = True,
                 topk: int = 1, seed: int = 0, rules: "WhisperRules | None" = None, path: str = "auto", kv32: bool = False) -> None:
        """``kv32``: the reference-accuracy form of the same step - fp32 memory in, cross and self K/V kept in fp32 (nothing is
        rounded when it is cached; pm_dec_attention_fused_kv32), everything else as in the throughput path, whose projections
        are fp32-exact already (bf16 weights x activations split into three bf16 terms) - graph-replayed like it."""
        if path not in ("auto", "launches", "persistent"):
            raise ValueError("greedy decode: path must be 'auto', 'launches' or 'persistent'")
        E = dec.token_embs.weight
        if E.dtype != torch.bfloat16 or not E.is_cuda:
            raise NotImplementedError("greedy decode: bf16 weights on a HIP device only (model.to(torch.bfloat16).cuda())")
        if memory is None:  # decoder-only language model (GPT-2): no cross-attention, nothing to attend to but itself
            B, S, d = prompt.shape[0], 0, E.shape[1]
        else:
            if memory.dtype != (torch.float32 if kv32 else torch.bfloat16) or memory.dim() != 3:
                raise ValueError("greedy decode: memory must be the encoder's (B, S, d) output, bf16 (fp32 with kv32=True)")
            B, S, d = memory.shape
        P = prompt.shape[1]
        if prompt.shape[0] != B or prompt.dtype != torch.int64 or P < 1:
            raise ValueError("greedy decode: prompt must be int64 (B, P >= 1)")
        V = E.shape[0]
        if int(prompt.min()) < 0 or int(prompt.max()) >= V:
            raise ValueError("greedy decode: prompt ids out of range")
        self.Ttot = P + n_new
        if self.Ttot > dec.pos_embs.shape[0]:
            raise ValueError(f"greedy decode: {self.Ttot} positions > max_seq_len {dec.pos_embs.shape[0]}")
        if B > 64:
            raise NotImplementedError("greedy decode: at most 64 sequences per call (shard larger batches)")
        dev = E.device
        ops.check_devices(E, memory, prompt if prompt.is_cuda else None)
        H = dec.layers[0].sa.n_heads
        inner = H * 64
        for i, layer in enumerate(dec.layers):  # the launch list below is built per layer from ONE geometry
            for name, att in (("sa", layer.sa), ("ca", layer.ca)):
                if att is not None and (att.head_dim != 64 or att.n_heads != H or att.n_heads * 64 != d):
                    raise NotImplementedError(f"greedy decode: layer {i} {name}: head_dim 64 with n_heads * 64 == d_model in every "
                                              f"layer only (got {att.n_heads} x {att.head_dim}, d_model {d})")
        hid_max = max(layer.mlp.linear1.out_features for layer in dec.layers)
        # One persistent launch for all layers of a step (csrc/decode_persist.hip) where its geometry rules hold; the
        # launch-per-stage list otherwise (and on request: tests compare the two)
        L = lib()
        nstep = 4 if d <= 512 else 8 if d <= 1024 else 10
        ksp_p = -(-(hid_max // 32) // (4 * nstep))
        acts = {layer.mlp.act_name for layer in dec.layers}
        hids = {layer.mlp.linear1.out_features for layer in dec.layers}
        persist_ok = (all(layer.pre_norm for layer in dec.layers) and len(acts) == 1 and len(hids) == 1 and d % 64 == 0 and d <= 1280
                      and hid_max % 32 == 0 and ksp_p <= 8 and S <= 2048 and self.Ttot <= 2048 and fused
                      and len({layer.ca is None for layer in dec.layers}) == 1 and hasattr(L, "pm_dec_layers")
                      and L.pm_dec_layers_grid() > 0)
        if path == "persistent" and not persist_ok:
            raise NotImplementedError("greedy decode: the persistent layer kernel needs pre-norm layers of one MLP width and activation, "
                                      "d_model % 64 == 0 <= 1280, memory and total length <= 2048")
        env_path = os.environ.get("PM_DEC_PATH")
        if path == "auto" and env_path in ("launches", "persistent"):
            path = env_path if (env_path == "launches" or persist_ok) else "launches"
        self.path = "persistent" if (path == "persistent" or (path == "auto" and persist_ok and PERSISTENT_BY_DEFAULT)) else "launches"
        persistent = self.path == "persistent"
        table = []
        # the attention block with the whole K stream in flight from the start (decode_persist.hip): opt-in, for A/B runs
        v2 = os.environ.get("PM_DEC_ATTN_V2", "0") != "0" and max(S, self.Ttot) <= 2048 and hasattr(L, "pm_dec_attention_fused_v2")  # measured slower (527 vs 461 us per step): off
        attn_fused = L.pm_dec_attention_fused_v2 if v2 else L.pm_dec_attention_fused
        self.kv32 = bool(kv32)
        kv_dt, kv_sz = (torch.float32, 4) if kv32 else (torch.bfloat16, 2)
        if kv32:
            if persistent or not fused or B * H > 256 or os.environ.get("PM_DEC_FUSE_SELF") == "0" or os.environ.get("PM_DEC_FUSE_CROSS") == "0":
                raise NotImplementedError("greedy decode: fp32 K/V caches run on the fused attention blocks (B * n_heads <= 256)")
            attn_fused = L.pm_dec_attention_fused_kv32
        self.B, self.P, self.n_steps = B, P, self.Ttot - 1
        Tmax = self.Ttot
        f32 = dict(dtype=torch.float32, device=dev)
        self.x = torch.empty(B, d, **f32)
        self.q = torch.empty(B, inner, **f32)
        self.att = torch.empty(B, inner, **f32)
        self.h = torch.empty(B, hid_max, **f32)  # widest MLP of the stack (mlp_ratio is free: transformer.py:77)
        self.pos = torch.zeros(1, dtype=torch.int32, device=dev)
        self.prompt = prompt.contiguous().to(dev)
        self.tok_cur = self.prompt[:, 0].clone()
        self.tokens = torch.zeros(B, self.Ttot, dtype=torch.int64, device=dev)
        self.tokens[:, :P] = self.prompt
        self.margins = torch.zeros(B, self.Ttot, **f32) if margins else None
        # d_model > 512: the final LayerNorm runs once as its own launch and the vocabulary projection without the
        # in-kernel LayerNorm, whose register budget would halve the feature tile (GPT-2 small: 101 -> ~55 us per step)
        self.split_final_norm = (d // 32 + 3) // 4 > 4
        tile = 64 if self.split_final_norm else L.pm_dec_argmax_tile(d)
        n_tiles = (V + tile - 1) // tile  # pm_dec_linear mode 2 leaves one (max, index) per tile and sequence
        self.ws_val = torch.empty(B, n_tiles, **f32)
        self.ws_idx = torch.empty(B, n_tiles, dtype=torch.int32, device=dev)
        pos_f32 = _f32(dec, "pos", dec.pos_embs)
        mem2 = memory.reshape(B * S, d) if memory is not None else None
        self._keep = [E, pos_f32, memory]  # tensors the launch list points into
        self.launches = []  # (fn, args): raw pointers only -> the loop has no per-step Python work beyond ctypes

        def add(fn, *args):
            self.launches.append((fn, args))

        # plain projections with a long K (fc2: K = 4 d) are split over workgroups along K: see pm_dec_linear_ksplit
        ks_min = int(os.environ.get("PM_DEC_KSPLIT_MINK", "1024"))
        self._ks_bufs, self._ks_cnts = [], []

        def dec_linear_ks(x, K, w, bias, resid, out, N, act=0):
            ksp = int(os.environ.get("PM_DEC_KSPLIT", "4"))
            ksp = max(2, min(ksp, 8, K // 32))
            nt, mt = (N + 15) // 16, (B + 15) // 16
            mt = 1 if mt <= 1 else 2 if mt == 2 else 4  # row tiles of the kernel instantiation
            ws = torch.empty(nt * ksp * mt * 256, **f32)
            cnt = torch.zeros(nt * 4, dtype=torch.int32, device=dev)  # one ticket per (feature tile, row tile)
            self._ks_bufs += [ws, cnt]
            self._ks_cnts.append(cnt)
            self._keep += [w, bias]
            add(L.pm_dec_linear_ksplit, x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), _ptr(bias), _ptr(resid),
                resid.stride(0) if resid is not None else 0, out.data_ptr(), out.stride(0), B, N, K, act, ksp,
                ws.data_ptr(), cnt.data_ptr(), None)

        def dec_linear(x, K, gamma, beta, eps, w, bias, resid, out, N, act=0, mode=0, kc=None, vc=None, ldo=None):
            if mode == 0 and gamma is None and K >= ks_min and N <= 4096 and int(os.environ.get("PM_DEC_KSPLIT", "4")) > 1:
                return dec_linear_ks(x, K, w, bias, resid, out, N, act)
            self._keep += [w, bias, gamma, beta]
            add(L.pm_dec_linear, x.data_ptr(), x.stride(0), _ptr(gamma), _ptr(beta), float(eps), w.data_ptr(), w.stride(0),
                _ptr(bias), _ptr(resid), resid.stride(0) if resid is not None else 0, _ptr(out),
                (out.stride(0) if out is not None else 0) if ldo is None else ldo, B, N, K, act, mode, _ptr(kc), _ptr(vc),
                inner, H, Tmax, self.pos.data_ptr(), self.ws_val.data_ptr(), self.ws_idx.data_ptr(), None)

        # x for position 0 comes from reset(); every later x row is written by the previous step's pm_dec_next_token
        self._embed0 = (L.pm_dec_embed, (self.tok_cur.data_ptr(), E.data_ptr(), pos_f32.data_ptr(), self.pos.data_ptr(),
                                         self.x.data_ptr(), B, d, V, None))
        self.ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        self.head_parts = self.x2 = self.fc2_parts = None
        cur, oth, pending = self.x, None, None  # the residual stream's buffer; the other one; sums the next fused block must add
        self.self_k, self.self_v, self.cross_kv, self._cross_w = [], [], [], []
        for layer in dec.layers:
            if not layer.pre_norm:
                raise NotImplementedError("greedy decode: pre-norm layers only (the post-norm GPT decodes through forward())")
            if (layer.ca is None) != (memory is None):
                raise ValueError("greedy decode: cross-attention layers need a memory, decoder-only layers must not get one")
            sa, ca, mlp = layer.sa, layer.ca, layer.mlp
            kc = torch.empty(B, H, Tmax, 64, dtype=kv_dt, device=dev)
            vc = torch.empty_like(kc)
            self.self_k.append(kc)
            self.self_v.append(vc)
            wqkv, bqkv = sa._pack("qkv")
            g, b = _f32(layer.sa_norm, "g", layer.sa_norm.weight), _f32(layer.sa_norm, "b", layer.sa_norm.bias)
            self._keep += [wqkv, bqkv, g, b]
            # the fused self block is one 512-thread workgroup per (sequence, head) that pulls the head's q/k/v weights
            # (3 * 64 * d * 2 B) through its CU: it wins while every workgroup has a CU to itself (B * H <= 256 on MI355X:
            # Whisper-base b = 32), beyond that the row-split projection + attention pair is faster (GPT-2 small b = 32,
            # B * H = 384: 705 -> 641 us per step)
            env_fs = os.environ.get("PM_DEC_FUSE_SELF")
            fuse_self = fused and (B * H <= 256 if env_fs is None else env_fs != "0")
            fuse_cross = fused and os.environ.get("PM_DEC_FUSE_CROSS", "1") != "0"
            # the chain of deferred sums (pm_dec_attention_chain): the self block leaves its output projection as per-head partial
            # sums that the cross block adds while it loads its row, fc2 leaves its K parts to the next layer's self block - one
            # launch and one ticket pass per layer less.  Needs both blocks fused (their workgroups own whole rows).
            chain = (fuse_self and fuse_cross and ca is not None and not persistent and not v2
                     and os.environ.get("PM_DEC_CHAIN", "1") != "0")
            if chain and self.head_parts is None:
                self.head_parts = torch.empty(B * H * d, **f32)
                self.x2 = torch.empty(B, d, **f32)
                oth = self.x2
            if persistent:
                bo = _f32(sa.out_proj, "b", sa.out_proj.bias)
                self._keep += [sa.out_proj.weight, bo]
                ent = DecLayer(sa_g=g.data_ptr(), sa_b=b.data_ptr(), w_qkv=wqkv.data_ptr(), b_qkv=_ptr(bqkv), kc=kc.data_ptr(),
                               vc=vc.data_ptr(), w_so=sa.out_proj.weight.data_ptr(), b_so=_ptr(bo), sa_eps=float(layer.sa_norm.eps))
                table.append(ent)
            elif chain:
                bo = _f32(sa.out_proj, "b", sa.out_proj.bias)
                self._keep += [sa.out_proj.weight, bo]
                n_in, p_in, ps_in, pr_in, pb_in = pending or (0, None, 0, 0, None)
                add(L.pm_dec_attention_chain, cur.data_ptr(), d, g.data_ptr(), b.data_ptr(), float(layer.sa_norm.eps),
                    wqkv.data_ptr(), _ptr(bqkv), kc.data_ptr(), vc.data_ptr(), H * Tmax * 64, Tmax * 64, 64, self.pos.data_ptr(),
                    0, Tmax, B, H, 1, int(kv32), p_in, n_in, ps_in, pr_in, pb_in, oth.data_ptr() if n_in else None,
                    sa.out_proj.weight.data_ptr(), self.head_parts.data_ptr(), None, None)
                if n_in:
                    cur, oth = oth, cur
                pending = (H, self.head_parts.data_ptr(), d, H * d, _ptr(bo))
            elif fuse_self:  # LN + q/k/v projection + cache append + attention in one launch per layer
                add(attn_fused, cur.data_ptr(), d, g.data_ptr(), b.data_ptr(), float(layer.sa_norm.eps),
                    wqkv.data_ptr(), _ptr(bqkv), kc.data_ptr(), vc.data_ptr(), H * Tmax * 64, Tmax * 64, 64, self.pos.data_ptr(),
                    0, Tmax, self.att.data_ptr(), B, H, 1, None)
            else:
                dec_linear(cur, d, g, b, layer.sa_norm.eps, wqkv, bqkv, None, self.q, 3 * inner, mode=1, kc=kc, vc=vc)
                add(L.pm_dec_attention, self.q.data_ptr(), kc.data_ptr(), vc.data_ptr(), H * Tmax * 64, Tmax * 64, 64,
                    self.pos.data_ptr(), 1, Tmax, self.att.data_ptr(), B, H, None)
            if not persistent and not chain:
                dec_linear(self.att, inner, None, None, 0.0, sa.out_proj.weight, _f32(sa.out_proj, "b", sa.out_proj.bias), cur,
                           cur, d)
            if ca is not None:
                # cross attention: K/V of the memory projected ONCE (the reference re-projects them on every call,
                # transformer.py:44-49), kept packed (B, S, [k | v]) in bf16
                wkv, bkv = ca._pack("kv")
                kv = self._project_memory(mem2, wkv, bkv)
                self.cross_kv.append(kv)
                self._cross_w.append((wkv, bkv))
                g, b = _f32(layer.ca_norm, "g", layer.ca_norm.weight), _f32(layer.ca_norm, "b", layer.ca_norm.bias)
                bq = _f32(ca.q_proj, "b", ca.q_proj.bias)
                self._keep += [g, b, bq]
                if persistent:
                    bo = _f32(ca.out_proj, "b", ca.out_proj.bias)
                    self._keep += [ca.q_proj.weight, ca.out_proj.weight, bo]
                    ent.ca_g, ent.ca_b, ent.ca_eps = g.data_ptr(), b.data_ptr(), float(layer.ca_norm.eps)
                    ent.w_q, ent.b_q, ent.cross_kv = ca.q_proj.weight.data_ptr(), _ptr(bq), kv.data_ptr()
                    ent.w_co, ent.b_co = ca.out_proj.weight.data_ptr(), _ptr(bo)
                elif chain:  # x = the stream + the self block's projection (heads in order) + its bias, formed while loading
                    n_in, p_in, ps_in, pr_in, pb_in = pending
                    add(L.pm_dec_attention_chain, cur.data_ptr(), d, g.data_ptr(), b.data_ptr(), float(layer.ca_norm.eps),
                        ca.q_proj.weight.data_ptr(), _ptr(bq), kv.data_ptr(), kv.data_ptr() + inner * kv_sz, S * 2 * inner, 64,
                        2 * inner, None, S, S, B, H, 0, int(kv32), p_in, n_in, ps_in, pr_in, pb_in, oth.data_ptr(), None, None,
                        self.att.data_ptr(), None)
                    cur, oth = oth, cur
                    pending = None
                elif fuse_cross:
                    add(attn_fused, cur.data_ptr(), d, g.data_ptr(), b.data_ptr(), float(layer.ca_norm.eps),
                        ca.q_proj.weight.data_ptr(), _ptr(bq), kv.data_ptr(), kv.data_ptr() + inner * kv_sz, S * 2 * inner, 64,
                        2 * inner, None, S, S, self.att.data_ptr(), B, H, 0, None)
                else:
                    dec_linear(cur, d, g, b, layer.ca_norm.eps, ca.q_proj.weight, bq, None, self.q, inner)
                    add(L.pm_dec_attention, self.q.data_ptr(), kv.data_ptr(), kv.data_ptr() + inner * 2, S * 2 * inner, 64,
                        2 * inner, None, S, S, self.att.data_ptr(), B, H, None)
                if not persistent:
                    dec_linear(self.att, inner, None, None, 0.0, ca.out_proj.weight, _f32(ca.out_proj, "b", ca.out_proj.bias), cur,
                               cur, d)
            g, b = _f32(layer.mlp_norm, "g", layer.mlp_norm.weight), _f32(layer.mlp_norm, "b", layer.mlp_norm.bias)
            if mlp.act_name not in ("gelu", "approximate_gelu"):
                raise NotImplementedError("greedy decode: GELU / tanh-GELU MLPs only")
            act_code = ops.ACT[mlp.act_name]
            hid = mlp.linear1.out_features
            if hid % 32 or mlp.linear2.in_features != hid or self.h[:, :hid].shape[1] != hid:
                raise NotImplementedError(f"greedy decode: MLP hidden width {hid} must be a multiple of 32 and fit the scratch row")
            if persistent:
                b1, b2 = _f32(mlp.linear1, "b", mlp.linear1.bias), _f32(mlp.linear2, "b", mlp.linear2.bias)
                self._keep += [g, b, mlp.linear1.weight, mlp.linear2.weight, b1, b2]
                ent.mlp_g, ent.mlp_b, ent.mlp_eps = g.data_ptr(), b.data_ptr(), float(layer.mlp_norm.eps)
                ent.w1, ent.b1, ent.w2, ent.b2 = mlp.linear1.weight.data_ptr(), _ptr(b1), mlp.linear2.weight.data_ptr(), _ptr(b2)
                continue
            dec_linear(cur, d, g, b, layer.mlp_norm.eps, mlp.linear1.weight, _f32(mlp.linear1, "b", mlp.linear1.bias), None,
                       self.h[:, :hid], hid, act=act_code)
            ksp = max(2, min(int(os.environ.get("PM_DEC_KSPLIT", "4")), 8, hid // 32))
            if chain and layer is not dec.layers[-1] and hid >= ks_min and int(os.environ.get("PM_DEC_KSPLIT", "4")) > 1:
                # fc2's K parts stay parts: the next layer's self block adds them (with the bias and this stream) while it loads
                if self.fc2_parts is None:
                    self.fc2_parts = torch.empty(8, B, d, **f32)
                b2 = _f32(mlp.linear2, "b", mlp.linear2.bias)
                self._keep += [mlp.linear2.weight, b2]
                add(L.pm_dec_linear_kparts, self.h.data_ptr(), self.h.stride(0), mlp.linear2.weight.data_ptr(),
                    mlp.linear2.weight.stride(0), self.fc2_parts.data_ptr(), d, B * d, B, d, hid, ksp, None)
                pending = (ksp, self.fc2_parts.data_ptr(), B * d, d, _ptr(b2))
            else:
                dec_linear(self.h[:, :hid], hid, None, None, 0.0, mlp.linear2.weight, _f32(mlp.linear2, "b", mlp.linear2.bias),
                           cur, cur, d)
        self.err = torch.zeros(1, dtype=torch.int32, device=dev)
        if persistent:
            import ctypes

            arr = (DecLayer * len(table))(*table)
            self.table = torch.frombuffer(bytearray(ctypes.string_at(ctypes.addressof(arr), ctypes.sizeof(arr))), dtype=torch.uint8).to(dev)
            mt = (B + 15) // 16
            self.ps_cnt = torch.zeros(len(table) * 24, dtype=torch.int32, device=dev)
            ws = torch.empty((d // 16) * mt * ksp_p * 256, **f32)
            tick = torch.zeros((d // 16) * mt, dtype=torch.int32, device=dev)
            self._ks_bufs += [ws, tick]
            self._ks_cnts.append(tick)
            add(L.pm_dec_layers, self.table.data_ptr(), len(table), B, d, H, S, Tmax, hid_max, ops.ACT[acts.pop()], ksp_p,
                self.pos.data_ptr(), self.x.data_ptr(), self.att.data_ptr(), self.h.data_ptr(), self.h.stride(0),
                self.ps_cnt.data_ptr(), ws.data_ptr(), tick.data_ptr(), self.err.data_ptr(), None)
        g, b = _f32(dec.norm, "g", dec.norm.weight), _f32(dec.norm, "b", dec.norm.bias)
        if not 1 <= topk <= 64:
            raise ValueError("greedy decode: topk must be in 1..64")
        self.topk = topk
        xl, gl, bl = cur, g, b  # (the chain leaves the stream in x or x2; the next step's row always goes to x)
        if self.split_final_norm:
            self.xn = torch.empty(B, d, **f32)
            self._keep += [g, b]
            add(L.pm_layernorm, cur.data_ptr(), d, 1, g.data_ptr(), b.data_ptr(), float(dec.norm.eps), self.xn.data_ptr(), d, 1,
                B, d, None)
            xl, gl, bl = self.xn, None, None
        if topk == 1 and rules is None:
            dec_linear(xl, d, gl, bl, dec.norm.eps, E, None, None, None, V, mode=2, ldo=0)
            # token choice + the next step's embedding row + position advance: one launch
            add(L.pm_dec_next_token, self.ws_val.data_ptr(), self.ws_idx.data_ptr(), n_tiles, self.pos.data_ptr(),
                self.prompt.data_ptr(), P, self.tok_cur.data_ptr(), self.tokens.data_ptr(), self.Ttot, _ptr(self.margins),
                E.data_ptr(), pos_f32.data_ptr(), self.x.data_ptr(), d, V, self.ticket.data_ptr(), B, None)
        else:  # top-k sampling on the device (text/generator.py:30-32): full logits of the last position, then the draw
            if margins:
                raise ValueError("greedy decode: margins are an arg-max diagnostic (topk == 1)")
            self.logits = torch.empty(B, V, **f32)
            dec_linear(xl, d, gl, bl, dec.norm.eps, E, None, None, self.logits, V, mode=0)
            if rules is not None:  # logit filters between the projection and the choice (k = 1 below = arg-max)
                i32 = dict(dtype=torch.int32, device=dev)
                sup, blk = torch.tensor(list(rules.suppress), **i32), torch.tensor(list(rules.blank), **i32)
                if not (0 <= rules.eot < rules.timestamp_begin < V) or any(not 0 <= int(i) < V for i in (*rules.suppress, *rules.blank)):
                    raise ValueError("WhisperRules: need 0 <= eot < timestamp_begin < vocab and listed ids inside the vocabulary")
                self._keep += [sup, blk]
                add(L.pm_dec_whisper_rules, self.logits.data_ptr(), self.logits.stride(0), V, self.tokens.data_ptr(), self.Ttot,
                    self.pos.data_ptr(), P, rules.eot, rules.no_timestamps, rules.timestamp_begin, rules.max_initial_timestamp,
                    sup.data_ptr() if sup.numel() else None, sup.numel(), blk.data_ptr() if blk.numel() else None, blk.numel(),
                    B, None)
            add(L.pm_dec_sample_topk, self.logits.data_ptr(), self.logits.stride(0), V, topk, int(seed) & (2**64 - 1),
                self.pos.data_ptr(), self.prompt.data_ptr(), P, self.tok_cur.data_ptr(), self.tokens.data_ptr(), self.Ttot,
                E.data_ptr(), pos_f32.data_ptr(), self.x.data_ptr(), d, self.ticket.data_ptr(), B, None)

    def _project_memory(self, mem2: Tensor, wkv: Tensor, bkv, out: Tensor | None = None) -> Tensor:
        """packed cross K/V of the memory rows: bf16 GEMM, or (kv32) the exact fp32 product of the fp32 memory with the
        bf16-valued weights."""
        if not self.kv32:
            return ops.linear(mem2, wkv, bkv, out=out)
        from ..transformer import derived

        w32 = derived(self, ("kv32w", wkv.data_ptr()), (wkv,), lambda: wkv.float())
        return ops.linear_f32(mem2, w32, bkv, out=out)

    def rebind(self, memory: Tensor, prompt: Tensor) -> None:
        """New clips, same geometry: re-project the cross K/V INTO the existing buffers and swap the prompt, so the
        captured graph (which holds raw pointers) stays valid."""
        assert (self.B, self.P) == tuple(prompt.shape)
        if memory is not None:
            B, S, d = memory.shape
            assert B == self.B and memory.dtype == (torch.float32 if self.kv32 else torch.bfloat16)
            mem2 = memory.reshape(B * S, d)
            for kv, (wkv, bkv) in zip(self.cross_kv, self._cross_w):
                assert kv.shape[0] == B * S
                self._project_memory(mem2, wkv, bkv, out=kv)
        self.prompt.copy_(prompt)
        self.tokens[:, : self.P] = self.prompt

    def step(self, log: dict | None = None) -> None:
        st = torch.cuda.current_stream().cuda_stream
        for fn, args in self.launches:
            if log is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            rc = fn(*args[:-1], st)
            if log is not None:
                e1.record()
                log.setdefault(fn.__name__, []).append((e0, e1, args))
            if rc:
                check(rc, fn.__name__)

    def reset(self) -> None:
        self.pos.zero_()
        self.ticket.zero_()
        self.err.zero_()
        if self.path == "persistent":
            self.ps_cnt.zero_()  # arrival counters count up by epochs of the position: zero with it
        for cnt in self._ks_cnts:  # the K-split tickets return to zero by themselves; this covers an aborted run
            cnt.zero_()
        self.tok_cur.copy_(self.prompt[:, 0])
        fn, args = self._embed0  # x[b] = emb[prompt[b, 0]] + pos[0]
        check(fn(*args[:-1], torch.cuda.current_stream().cuda_stream), "pm_dec_embed")

    def run(self, graph: bool = True) -> Tensor:
        if not graph:
            self.reset()
            for _ in range(self.n_steps):
                self.step()
            return self.tokens
        if getattr(self, "_graph", None) is None:
            self.reset()
            self.step()  # eager warm-up: loads every kernel before capture
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.step()
            self._graph = g
        self.reset()
        for _ in range(self.n_steps):
            self._graph.replay()
        return self.tokens


    def check(self) -> None:
        """Raise if a hand-off inside the persistent step kernel gave up (one read-back: call it where the tokens are consumed)."""
        if self.path == "persistent" and int(self.err.item()) != 0:
            raise RuntimeError("greedy decode: a hand-off of the persistent decode-step kernel timed out (not every workgroup of "
                               "the launch was resident, or the device is shared); tokens of this run are invalid - re-run with "
                               "path='launches'")


@torch.no_grad()
def greedy_decode(dec, memory: Tensor, prompt: Tensor, n_new: int, *, graph: bool = True, margins: bool = False,
                  fused: bool = True, topk: int = 1, seed: int = 0, rules: "WhisperRules | None" = None, path: str = "auto",
                  kv32: bool = False):
    """tokens (B, P + n_new) int64 [and per-position diagnostic margins].  fused=False uses the unfused
    projection + attention launches (same arithmetic, 2 more launches per layer); topk > 1 samples each token from the
    softmax over the k largest logits on the device (same seed -> same ids); path: "persistent" (all layers of a step in
    one launch), "launches" (a launch per stage) or "auto"."""
    st = GreedyDecoder(dec, memory, prompt, n_new, margins, fused, topk, seed, rules, path, kv32)
    toks = st.run(graph)
    st.check()
    return (toks, st.margins) if margins else toks


@torch.no_grad()
def greedy_exact(dec, memory: Tensor | None, prompt: Tensor, n_new: int, *, margins: bool = False):
    """KV-cached greedy decoding in fp32 END TO END for a decoder whose parameters are fp32: fp32 weights, fp32 activations,
    fp32 self and cross K/V (nothing rounded to bf16 anywhere), one token per step through pm_linear_f32 /
    pm_attention_generic_f32 / pm_layernorm.  This is the path whose ids are compared bit for bit with the reference's fp32
    full-prefix loop (text/generator.py:23-35 semantics; tests/golden/whisper.npz greedy_*_224): same algebra as the
    reference, cached instead of recomputed (the single-token step attends WITHOUT a causal flag: SURVEY.md F3).
    ~14 eager launches per layer and step - a reference-accuracy mode, not the throughput path (that is GreedyDecoder on a
    bf16 model).  It is also the KV-cached decode of what GreedyDecoder's step kernels do not cover: post-norm stacks (GPT)
    and heads other than 64 wide / n_heads * head_dim != d_model (any head_dim % 8 == 0 up to 128).  Returns tokens (B, P + n_new) [and the top-1 minus top-2 logit of every generated position]."""
    from ..transformer import MHA

    E = dec.token_embs.weight
    if E.dtype != torch.float32 or not E.is_cuda:
        raise NotImplementedError("greedy_exact: fp32 parameters on a HIP device (bf16 models: GreedyDecoder, or Whisper.generate(exact=True))")
    B, P = prompt.shape
    V, d = E.shape
    Ttot = P + n_new
    if Ttot > dec.pos_embs.shape[0]:
        raise ValueError(f"greedy decode: {Ttot} positions > max_seq_len {dec.pos_embs.shape[0]}")
    dev = E.device
    prompt = prompt.to(dev)
    tokens = torch.zeros(B, Ttot, dtype=torch.int64, device=dev)
    tokens[:, :P] = prompt
    marg = torch.zeros(B, Ttot, dtype=torch.float32, device=dev) if margins else None
    pos = dec.pos_embs.float()
    layers = list(dec.layers)
    for layer in layers:
        if type(layer.sa) is not MHA or layer.sa.head_dim % 8 or layer.sa.head_dim > 128:
            raise NotImplementedError("greedy_exact: plain MHA layers with head_dim % 8 == 0 (<= 128)")
    cross, caches = [], []
    for layer in layers:
        inner = layer.sa.n_heads * layer.sa.head_dim
        caches.append((torch.empty(B, Ttot, inner, dtype=torch.float32, device=dev), torch.empty(B, Ttot, inner, dtype=torch.float32, device=dev)))
        if layer.ca is not None:
            if memory is None:
                raise ValueError("greedy_exact: cross-attention layers need a memory")
            w, b = layer.ca._pack32("kv")
            mem = memory.float()
            ci = layer.ca.n_heads * layer.ca.head_dim
            kv = ops.linear_f32(mem.reshape(-1, mem.shape[-1]), w, b).view(B, mem.shape[1], 2 * ci)
            cross.append((kv[..., :ci], kv[..., ci:]))
        else:
            cross.append(None)
    for t in range(Ttot - 1):
        x = ops.embed_tokens(tokens[:, t : t + 1], E, pos, pos0=t).view(B, d)  # f32 rows
        for layer, (kc, vc), xkv in zip(layers, caches, cross):
            # pre-norm: x + f(norm(x)); post-norm (GPT, text/gpt.py:23): norm(x + f(x)) - transformer.py:96-105
            pre = layer.pre_norm
            sa = layer.sa
            inner = sa.n_heads * sa.head_dim
            w, b = sa._pack32("qkv")
            qkv = ops.linear_f32(layer.sa_norm(x) if pre else x, w, b)
            kc[:, t] = qkv[:, inner : 2 * inner]
            vc[:, t] = qkv[:, 2 * inner :]
            a = ops.attention_f32(qkv[:, :inner].unsqueeze(1), kc[:, : t + 1], vc[:, : t + 1], sa.n_heads)
            x = ops.linear_f32(a.view(B, inner), sa.out_proj.weight, sa.out_proj.bias, resid=x)
            if not pre:
                x = layer.sa_norm(x)
            if xkv is not None:
                ca = layer.ca
                q = ops.linear_f32(layer.ca_norm(x) if pre else x, ca.q_proj.weight, ca.q_proj.bias)
                a = ops.attention_f32(q.unsqueeze(1), xkv[0], xkv[1], ca.n_heads)
                x = ops.linear_f32(a.view(B, -1), ca.out_proj.weight, ca.out_proj.bias, resid=x)
                if not pre:
                    x = layer.ca_norm(x)
            mlp = layer.mlp
            h = ops.linear_f32(layer.mlp_norm(x) if pre else x, mlp.linear1.weight, mlp.linear1.bias, act=mlp.act_name)
            x = ops.linear_f32(h, mlp.linear2.weight, mlp.linear2.bias, resid=x)
            if not pre:
                x = layer.mlp_norm(x)
        logits = ops.linear_f32(dec.norm(x) if getattr(dec, "norm", None) is not None else x, E)  # (B, V) f32
        if t + 1 < P:
            tokens[:, t + 1] = prompt[:, t + 1]
        else:
            if margins:
                top2 = logits.topk(2, -1)
                marg[:, t + 1] = top2.values[:, 0] - top2.values[:, 1]
                tokens[:, t + 1] = top2.indices[:, 0]
            else:
                tokens[:, t + 1] = logits.argmax(-1)
    return (tokens, marg) if margins else tokens
