#!/usr/bin/env python3
"""Lane-level numpy emulation of the two MFMA kernels' index math (CPU, no GPU needed).

Operand / result layouts follow /opt/skills/guides/cdna_hip_programming.md section 3:
  v_mfma_f32_16x16x4_f32 : lane l gives A[i=l&15][k=l>>4], B[k=l>>4][j=l&15];
                           D reg r of lane l = D[row=(l>>4)*4+r][col=l&15]
  v_mfma_f32_32x32x2_f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
                           D reg r of lane l = D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]
It replays nce_tile<D,PASS2> and gemm_nt_kernel<D> exactly as written in
selfrec_amd/csrc/{losses,eval}.hip and compares with dense numpy.  A layout slip in the
kernels shows up here before any GPU time is spent.
"""
import numpy as np


def mfma16(a, b, c):
    """a,b: (64,) per-lane scalars; c: (64,4). Returns new c."""
    A = np.zeros((16, 4)); B = np.zeros((4, 16))
    for l in range(64):
        A[l & 15, l >> 4] = a[l]
        B[l >> 4, l & 15] = b[l]
    Dm = A @ B
    out = c.copy()
    for l in range(64):
        for r in range(4):
            out[l, r] += Dm[(l >> 4) * 4 + r, l & 15]
    return out


def mfma32(a, b, c):
    A = np.zeros((32, 2)); B = np.zeros((2, 32))
    for l in range(64):
        A[l & 31, l >> 5] = a[l]
        B[l >> 5, l & 31] = b[l]
    Dm = A @ B
    out = c.copy()
    for l in range(64):
        for r in range(16):
            out[l, r] += Dm[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return out


def nce_tile_wave(Q, K, q0, kb, ke, n, inv_tau, invl=None):
    """One wave of nce_tile: returns O (16, D) partial and l (16,) partial."""
    D = Q.shape[1]; DQ = D // 4; NT = D // 16
    lanes = np.arange(64); c16 = lanes & 15; g = lanes >> 4
    qreg = np.stack([Q[q0 + c16[l], g[l] * DQ:(g[l] + 1) * DQ] for l in range(64)])
    O = np.zeros((NT, 64, 4)); lsum = np.zeros(64)
    for j0 in range(kb, ke, 32):
        for h in range(2):
            kreg = np.stack([K[j0 + 16 * h + c16[l], g[l] * DQ:(g[l] + 1) * DQ] for l in range(64)])
            acc = np.zeros((64, 4))
            for s in range(DQ):
                acc = mfma16(kreg[:, s], qreg[:, s], acc)
            wgt = np.zeros((64, 4))
            for l in range(64):
                for r in range(4):
                    key = j0 + 16 * h + 4 * g[l] + r
                    e = np.exp(acc[l, r] * inv_tau - inv_tau)
                    if invl is not None:
                        e *= invl[min(key, len(invl) - 1)]
                    wgt[l, r] = e if key < n else 0.0
                    lsum[l] += wgt[l, r]
            for s in range(4):
                vv = np.stack([K[j0 + 16 * h + 4 * g[l] + s, c16[l] * NT:(c16[l] + 1) * NT] for l in range(64)])
                for t in range(NT):
                    O[t] = mfma16(wgt[:, s], vv[:, t], O[t])
    out = np.zeros((16, D))
    for l in range(64):
        for r in range(4):
            for t in range(NT):
                out[4 * g[l] + r, NT * c16[l] + t] = O[t][l, r]
    lq = np.zeros(16)
    for l in range(64):
        lq[c16[l]] += lsum[l]
    return out, lq


def check_nce(D=64, n=45, np_=64, tau=0.2, seed=0):
    rng = np.random.default_rng(seed)
    v1 = np.zeros((np_, D)); v2 = np.zeros((np_, D))
    a = rng.standard_normal((n, D)); b = rng.standard_normal((n, D))
    v1[:n] = a / np.linalg.norm(a, axis=1, keepdims=True)
    v2[:n] = b / np.linalg.norm(b, axis=1, keepdims=True)
    it = 1 / tau
    S = v1[:n] @ v2[:n].T * it
    W = np.exp(S - it)
    O_ref = W @ v2[:n]; l_ref = W.sum(1)
    O = np.zeros((np_, D)); l = np.zeros(np_)
    for q0 in range(0, np_, 16):
        o, lq = nce_tile_wave(v1, v2, q0, 0, np_, n, it)
        O[q0:q0 + 16] = o; l[q0:q0 + 16] = lq
    assert np.allclose(O[:n], O_ref), "pass1 O mismatch"
    assert np.allclose(l[:n], l_ref), "pass1 l mismatch"
    invl = np.zeros(np_); invl[:n] = 1 / l_ref
    O2_ref = (W / l_ref[:, None]).T @ v1[:n]
    O2 = np.zeros((np_, D))
    for q0 in range(0, np_, 16):
        o, _ = nce_tile_wave(v2, v1, q0, 0, np_, n, it, invl)
        O2[q0:q0 + 16] = o
    assert np.allclose(O2[:n], O2_ref), "pass2 O mismatch"
    print(f"nce_tile emulation OK (D={D}, n={n})")


def gemm_wave(A, B, m0, n0):
    D = A.shape[1]; DH = D // 2
    lanes = np.arange(64); r32 = lanes & 31; h = lanes >> 5
    m, n = A.shape[0], B.shape[0]
    a = np.stack([A[min(m0 + r32[l], m - 1), h[l] * DH:(h[l] + 1) * DH] for l in range(64)])
    b = np.stack([B[min(n0 + r32[l], n - 1), h[l] * DH:(h[l] + 1) * DH] for l in range(64)])
    acc = np.zeros((64, 16))
    for s in range(DH):
        acc = mfma32(a[:, s], b[:, s], acc)
    C = {}
    for l in range(64):
        for t in range(16):
            row = m0 + (t & 3) + 8 * (t >> 2) + 4 * h[l]; col = n0 + r32[l]
            if row < m and col < n:
                C[(row, col)] = acc[l, t]
    return C


def check_gemm(D=64, m=40, n=70, seed=1):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, D)); B = rng.standard_normal((n, D))   # asymmetric operands
    ref = A @ B.T
    C = np.full((m, n), np.nan)
    for m0 in range(0, m, 32):
        for n0 in range(0, n, 32):
            for (r, c), v in gemm_wave(A, B, m0, n0).items():
                C[r, c] = v
    assert np.allclose(C, ref), "gemm mismatch"
    print(f"gemm_nt emulation OK (D={D}, m={m}, n={n})")


if __name__ == "__main__":
    check_nce(64, 45, 64)
    check_nce(128, 70, 128)
    check_gemm(64)
    check_gemm(128, 33, 65)


# ---- split-bf16 InfoNCE tile (nce_tile_bf16): layout check with exact arithmetic (lo parts = 0) ----
def mfma16x32(a, b, c):
    """a, b: (64, 8) per-lane fragments; A[i=l&15][k=8*(l>>4)+e], B[k=8*(l>>4)+e][j=l&15]."""
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for l in range(64):
        for e in range(8):
            A[l & 15, 8 * (l >> 4) + e] = a[l, e]
            B[8 * (l >> 4) + e, l & 15] = b[l, e]
    Dm = A @ B
    out = c.copy()
    for l in range(64):
        for r in range(4):
            out[l, r] += Dm[(l >> 4) * 4 + r, l & 15]
    return out


def vt_image(V):
    """nce_prep's key-blocked transposed image: vt[blk][col][8*((k&15)>>2) + 4*(k>>4) + (k&3)] = V[32 blk + k][col]."""
    n, D = V.shape
    vt = np.zeros((n // 32, D, 32))
    for i in range(n):
        k = i & 31
        vt[i >> 5, :, 8 * ((k & 15) >> 2) + 4 * (k >> 4) + (k & 3)] = V[i]
    return vt


def nce_tile_bf16_wave(Q, K, q0, kb, ke, n, inv_tau, invl=None):
    D = Q.shape[1]; NT = D // 16; KS = D // 32; DG = D // 4
    lanes = np.arange(64); c16 = lanes & 15; g = lanes >> 4
    vt = vt_image(K)
    qf = [np.stack([Q[q0 + c16[l], DG * g[l] + 8 * s: DG * g[l] + 8 * s + 8] for l in range(64)]) for s in range(KS)]
    O = np.zeros((NT, 64, 4)); lsum = np.zeros(64)
    for j0 in range(kb, ke, 32):
        acc = []
        for h in range(2):
            a = np.zeros((64, 4))
            for s in range(KS):
                kf = np.stack([K[j0 + 16 * h + c16[l], DG * g[l] + 8 * s: DG * g[l] + 8 * s + 8] for l in range(64)])
                a = mfma16x32(kf, qf[s], a)
            acc.append(a)
        p = np.zeros((64, 8))
        for l in range(64):
            for h in range(2):
                for r in range(4):
                    key = j0 + 16 * h + 4 * g[l] + r
                    e = np.exp(acc[h][l, r] * inv_tau - inv_tau)
                    if invl is not None:
                        e *= invl[min(key, len(invl) - 1)]
                    wt = e if key < n else 0.0
                    lsum[l] += wt
                    p[l, 4 * h + r] = wt
        for t in range(NT):
            vf = np.stack([vt[j0 >> 5, NT * c16[l] + t, 8 * g[l]: 8 * g[l] + 8] for l in range(64)])
            O[t] = mfma16x32(p, vf, O[t])
    out = np.zeros((16, D))
    for l in range(64):
        for r in range(4):
            for t in range(NT):
                out[4 * g[l] + r, NT * c16[l] + t] = O[t][l, r]
    lq = np.zeros(16)
    for l in range(64):
        lq[c16[l]] += lsum[l]
    return out, lq


def check_nce_bf16(D=64, n=45, np_=64, tau=0.2, seed=3):
    rng = np.random.default_rng(seed)
    v1 = np.zeros((np_, D)); v2 = np.zeros((np_, D))
    a = rng.standard_normal((n, D)); b = rng.standard_normal((n, D))
    v1[:n] = a / np.linalg.norm(a, axis=1, keepdims=True)
    v2[:n] = b / np.linalg.norm(b, axis=1, keepdims=True)
    it = 1 / tau
    W = np.exp(v1[:n] @ v2[:n].T * it - it)
    O = np.zeros((np_, D)); l = np.zeros(np_)
    for q0 in range(0, np_, 16):
        O[q0:q0 + 16], l[q0:q0 + 16] = nce_tile_bf16_wave(v1, v2, q0, 0, np_, n, it)
    assert np.allclose(O[:n], W @ v2[:n]) and np.allclose(l[:n], W.sum(1)), "bf16 pass1 mismatch"
    invl = np.zeros(np_); invl[:n] = 1 / W.sum(1)
    O2 = np.zeros((np_, D))
    for q0 in range(0, np_, 16):
        O2[q0:q0 + 16], _ = nce_tile_bf16_wave(v2, v1, q0, 0, np_, n, it, invl)
    assert np.allclose(O2[:n], (W / W.sum(1)[:, None]).T @ v1[:n]), "bf16 pass2 mismatch"
    print(f"nce_tile_bf16 emulation OK (D={D}, n={n})")


if __name__ == "__main__":
    check_nce_bf16(64, 45, 64)
    check_nce_bf16(128, 100, 128)
