"""Test helper: execute a native engine's plan (gd_engine_plan_json) on the CPU with torch.

Not product code and not the oracle: it emulates, op by op, what the HIP kernels are specified to do
(csrc/kernels/*.hip), reading/writing one flat arena exactly as the engine does (address =
offset * batch), with the engine's own packed weight slots.  Running it against the oracle U-Net
checks the planner (op order, buffer lifetimes/aliasing, weight names and packing, concat handling,
FIR composition) on a machine without a GPU.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

OP_FOURIER, OP_GEMM, OP_GN_STATS, OP_GN_APPLY, OP_FIR_UP, OP_FIR_DOWN, OP_SOFTMAX, OP_AFFINE_PAD, OP_SCALE_ROWS, OP_GN_GENERIC = range(10)
EXT_NONE, EXT_X_IN, EXT_LABELS, EXT_EPS_OUT = range(4)


class PlanInterp(object):
    def __init__(self, engine, dtype=torch.float64):
        self.plan = engine.plan()
        self.dtype = dtype
        self.slots = [torch.as_tensor(engine.packed_weight(s["name"]).astype(np.float64)).to(dtype).reshape(s["rows"], s["cols"])
                      for s in self.plan["slots"]]

    def _flat(self, tref, batch):
        """1-D view starting at the tensor's base + column offset (like a device pointer)."""
        t, col = tref
        if t < 0:
            return None
        T = self.plan["tensors"][t]
        if T["ext"] == EXT_NONE:
            base = self.arena[T["offset"] * batch:]
        else:
            base = self.ext[T["ext"]]
        return base[col:]

    def run(self, x_in, labels):
        """x_in (B,H,W,CP) padded input, labels (B,) -> eps (B,H,W,in_ch)."""
        B = x_in.shape[0]
        P = self.plan
        self.arena = torch.full((P["arena_per_image"] * B,), float("nan"), dtype=self.dtype)
        t_out = [t for t in P["tensors"] if t["ext"] == EXT_EPS_OUT][0]
        out = torch.full((B * t_out["per_image"],), float("nan"), dtype=self.dtype)
        self.ext = {EXT_X_IN: x_in.to(self.dtype).reshape(-1), EXT_LABELS: labels.to(self.dtype).reshape(-1), EXT_EPS_OUT: out}
        for op in P["ops"]:
            getattr(self, "_op%d" % op["kind"])(op, B)
        return out.reshape(B, t_out["H"], t_out["W"], t_out["C"])

    # ---- helpers ----
    @staticmethod
    def _rows(flat, nrows, ld, ncols):
        """(nrows, ncols) strided view: element (r, c) at flat[r*ld + c]."""
        return torch.as_strided(flat, (nrows, ncols), (ld, 1))

    def _gather_a(self, op, B):
        """Implicit im2col: returns A (M, K) with k = (dy*KW+dx)*Cin + c, and M."""
        H, W, OH, OW, KH, KW, st, pad = (op[k] for k in ("H", "W", "OH", "OW", "KH", "KW", "stride", "pad"))
        C0, C1 = op["C0"], op["C1"]
        nimg = 1 if op["z_per_image"] else B
        srcs = []
        for tref, C, ld in ((op["a0"], C0, op["lda0"]), (op["a1"], C1, op["lda1"])):
            if C == 0:
                continue
            flat = self._flat(tref, B)
            srcs.append(self._rows(flat, nimg * H * W, ld, C).reshape(nimg, H, W, C))
        x = torch.cat(srcs, dim=-1) if len(srcs) > 1 else srcs[0]
        if op["in_act"]:
            x = x * torch.sigmoid(x)
        xp = F.pad(x, (0, 0, pad, pad, pad, pad))
        cols = []
        for dy in range(KH):
            for dx in range(KW):
                cols.append(xp[:, dy:dy + st * (OH - 1) + 1:st, dx:dx + st * (OW - 1) + 1:st, :])
        A = torch.cat(cols, dim=-1).reshape(nimg * OH * OW, KH * KW * (C0 + C1))
        return A

    # ---- ops ----
    def _op0(self, op, B):  # fourier
        labels = self._flat(op["a0"], B)[:B]
        W = self.slots[op["w_slot"]].reshape(-1)
        E = op["N"]
        if op["act"]:  # positional embedding: sin / cos(label * frequency)
            xp = labels[:, None] * W[None, :]
        else:
            xp = torch.log(labels)[:, None] * W[None, :] * 2 * math.pi
        o = self._rows(self._flat(op["out"], B), B, 2 * E, 2 * E)
        o[:, :E] = torch.sin(xp)
        o[:, E:] = torch.cos(xp)

    def _op1(self, op, B):  # gemm
        N, K = op["N"], op["K"]
        nz = B if op["z_per_image"] else 1
        Mz = op["m_per_image"] if op["z_per_image"] else op["m_per_image"] * B
        for z in range(nz):
            o = dict(op)
            if op["z_per_image"]:
                # shift base pointers by z * stride by slicing flats
                if op.get("a_slot", -1) >= 0:
                    a0 = self.slots[op["a_slot"]].reshape(-1)[op["a_off"]:]
                else:
                    a0 = self._flat(op["a0"], B)[z * op["sA"]:]
                H, W = op["H"], op["W"]
                x = self._rows(a0, H * W, op["lda0"], op["C0"]).reshape(1, H, W, op["C0"])
                A = x.reshape(Mz, K)
                bflat = self._flat(op["bten"], B)[z * op["sB"]:]
            else:
                A = self._gather_a(op, B)
                bflat = None if op["w_slot"] >= 0 else self._flat(op["bten"], B)
            if op["w_slot"] >= 0:
                Bm = self.slots[op["w_slot"]]
                assert Bm.shape[0] >= N and Bm.shape[1] == K, (op["label"], Bm.shape, N, K)
                Bm = Bm[:N]  # a GEMM may use the leading rows of a shared slot (q,k of the qkv slot)
            elif op["b_kmajor"]:
                Bm = self._rows(bflat, K, op["ldb"], N).t()
            else:
                Bm = self._rows(bflat, N, op["ldb"], K)
            acc = op["alpha"] * (A @ Bm.t())
            if op["bias_slot"] >= 0:
                acc = acc + self.slots[op["bias_slot"]].reshape(-1)[op["bias_off"]:op["bias_off"] + N]
            if op["rowbias"][0] >= 0:
                rb = self._rows(self._flat(op["rowbias"], B), B, op["ld_rowbias"], N)
                acc = acc + rb.repeat_interleave(op["m_per_image"], dim=0)
            if op["res"][0] >= 0:
                acc = acc + self._rows(self._flat(op["res"], B)[z * op["sR"]:], Mz, op["ldres"], N)
            oflat = self._flat(op["out"], B)[z * op["sC"]:]
            res_out = acc * op["out_scale"]
            self._rows(oflat, Mz, op["ldc"], N)[:] = res_out
            if op["qstats"][0] >= 0:
                # fused GroupNorm statistics (kernels/qstats.h): per image, per slice (32 rows; the whole image when it has
                # 16 rows), per 4 consecutive channels: (sum, sum of squares)
                rpi, S, Q = op["m_per_image"], op["qs_slices"], N // 4
                rows_per_slice = 32 if rpi % 32 == 0 else rpi
                assert S == rpi // rows_per_slice
                nb = 1 if op["z_per_image"] else B  # a batched product writes the slices of image z at z * sQ
                assert not op["z_per_image"] or op["sQ"] == S * Q * 2
                v = res_out.reshape(nb, S, rows_per_slice, Q, 4)
                qs = self._rows(self._flat(op["qstats"], B)[z * op["sQ"]:], nb * S * Q, 2, 2).reshape(nb, S, Q, 2)
                qs[..., 0] = v.sum(dim=(2, 4))
                qs[..., 1] = (v * v).sum(dim=(2, 4))

    def _gn_src(self, op, B):
        HW = op["H"] * op["W"]
        srcs = []
        for tref, C in ((op["a0"], op["C0"]), (op["a1"], op["C1"])):
            if C:
                srcs.append(self._rows(self._flat(tref, B), B * HW, C, C).reshape(B, HW, C))
        return torch.cat(srcs, dim=-1) if len(srcs) > 1 else srcs[0]

    RT_SLICES = 2  # the emulation's stand-in for the run-time slice count of the stand-alone statistics kernel

    def _op2(self, op, B):  # gn stats (stand-alone): quad partial sums qs[b][slice][c/4] = (sum, sumsq)
        HW, C = op["H"] * op["W"], op["C0"]
        x = self._rows(self._flat(op["a0"], B), B * HW, C, C).reshape(B, HW, C)
        S, Q = self.RT_SLICES, C // 4
        per = (HW + S - 1) // S
        qs = self._rows(self._flat(op["stats"], B), B * S * Q, 2, 2).reshape(B, S, Q, 2)
        for sl in range(S):
            v = x[:, sl * per:min(HW, (sl + 1) * per)].reshape(B, -1, Q, 4)
            qs[:, sl, :, 0] = v.sum(dim=(1, 3))
            qs[:, sl, :, 1] = (v * v).sum(dim=(1, 3))

    def _op3(self, op, B):  # gn apply: statistics from the quad sums of up to two sources, then normalise (+ swish)
        x = self._gn_src(op, B)
        G = op["groups"]
        Bn, HW, C = x.shape
        parts = []
        for tref, Cs, S in ((op["stats"], op["C0"], op["S0"]), (op["stats1"], op["C1"], op["S1"])):
            if Cs == 0:
                continue
            S = S if S > 0 else self.RT_SLICES
            Q = Cs // 4
            qs = self._rows(self._flat(tref, B), B * S * Q, 2, 2).reshape(B, S, Q, 2)
            parts.append(qs.sum(dim=1))  # (B, Q, 2)
        q = torch.cat(parts, dim=1)  # channel-concat order
        g = q.reshape(B, G, (C // 4) // G, 2).sum(dim=2)
        n = HW * (C // G)
        mean = g[..., 0] / n
        var = g[..., 1] / n - mean * mean
        rstd = torch.rsqrt(var + 1e-6)
        xg = x.reshape(Bn, HW, G, C // G)
        y = ((xg - mean[:, None, :, None]) * rstd[:, None, :, None]).reshape(Bn, HW, C)
        y = y * self.slots[op["gamma_slot"]].reshape(-1) + self.slots[op["beta_slot"]].reshape(-1)
        if op["act"]:
            y = y * torch.sigmoid(y)
        self._rows(self._flat(op["out"], B), B * HW, C, C)[:] = y.reshape(B * HW, C)

    def _op9(self, op, B):  # generic GroupNorm (groups that are not whole float4 columns)
        x = self._gn_src(op, B)
        G = op["groups"]
        Bn, HW, C = x.shape
        xg = x.reshape(Bn, HW, G, C // G)
        mean = xg.mean(dim=(1, 3), keepdim=True)
        var = (xg * xg).mean(dim=(1, 3), keepdim=True) - mean * mean
        y = ((xg - mean) * torch.rsqrt(var + 1e-6)).reshape(Bn, HW, C)
        y = y * self.slots[op["gamma_slot"]].reshape(-1) + self.slots[op["beta_slot"]].reshape(-1)
        if op["act"]:
            y = y * torch.sigmoid(y)
        self._rows(self._flat(op["out"], B), B * HW, C, C)[:] = y.reshape(B * HW, C)

    def _fir(self, op, B, up):
        H, W, C = op["H"], op["W"], op["C0"]
        x = self._rows(self._flat(op["a0"], B), B * H * W, C, C).reshape(B, H, W, C)
        k = torch.tensor([1.0, 3.0, 3.0, 1.0], dtype=self.dtype)
        if op["act"]:  # fir=False: nearest-neighbour up / 2x2 mean down
            if up:
                y = x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
                OH, OW = 2 * H, 2 * W
            else:
                y = x.reshape(B, H // 2, 2, W // 2, 2, C).mean(dim=(2, 4))
                OH, OW = H // 2, W // 2
        elif up:
            # y[2m] = (x[m-1] + 3 x[m]) / 4 ; y[2m+1] = (3 x[m] + x[m+1]) / 4, per axis
            def up1(t, dim):
                n = t.shape[dim]
                tp = F.pad(t.movedim(dim, -1), (1, 1)).movedim(-1, dim)
                a = (tp.narrow(dim, 0, n) + 3 * tp.narrow(dim, 1, n)) / 4
                b = (3 * tp.narrow(dim, 1, n) + tp.narrow(dim, 2, n)) / 4
                return torch.stack([a, b], dim=dim + 1).reshape(*t.shape[:dim], 2 * n, *t.shape[dim + 1:])
            y = up1(up1(x, 1), 2)
            OH, OW = 2 * H, 2 * W
        else:
            def dn1(t, dim):
                n = t.shape[dim]
                tp = F.pad(t.movedim(dim, -1), (1, 1)).movedim(-1, dim)
                idx = torch.arange(0, n, 2)
                return sum(k[j] / 8 * tp.index_select(dim, idx + j) for j in range(4))
            y = dn1(dn1(x, 1), 2)
            OH, OW = H // 2, W // 2
        self._rows(self._flat(op["out"], B), B * OH * OW, C, C)[:] = y.reshape(B * OH * OW, C)

    def _op4(self, op, B):
        self._fir(op, B, True)

    def _op5(self, op, B):
        self._fir(op, B, False)

    def _op10(self, op, B):  # FIR of conv_downsample_2d at full resolution: (H, W) -> (H + 1, W + 1), pads (2, 2)
        H, W, C = op["H"], op["W"], op["C0"]
        x = self._rows(self._flat(op["a0"], B), B * H * W, C, C).reshape(B, H, W, C)
        k = torch.tensor([1.0, 3.0, 3.0, 1.0], dtype=self.dtype) / 8
        xp = F.pad(x.permute(0, 3, 1, 2), (2, 2, 2, 2))
        y = sum(k[i] * k[j] * xp[:, :, i:i + H + 1, j:j + W + 1] for i in range(4) for j in range(4)).permute(0, 2, 3, 1)
        self._rows(self._flat(op["out"], B), B * (H + 1) * (W + 1), C, C)[:] = y.reshape(B * (H + 1) * (W + 1), C)

    def _op6(self, op, B):  # softmax in place
        rows, cols = B * op["rows_per_image"], op["cols"]
        v = self._rows(self._flat(op["out"], B), rows, cols, cols)
        v[:] = torch.softmax(v, dim=-1)

    def _op7(self, op, B):  # affine pad
        pix = B * op["H"] * op["W"]
        x = self._rows(self._flat(op["a0"], B), pix, op["lda0"], op["cin"])
        o = self._rows(self._flat(op["out"], B), pix, op["cpad"], op["cpad"])
        o[:] = 0
        o[:, :op["cin"]] = x * op["mul"] + op["add"]

    def _op8(self, op, B):  # scale rows
        per = op["m_per_image"]
        v = self._rows(self._flat(op["out"], B), B, per, per)
        v[:] = v / self._flat(op["a0"], B)[:B, None]
