"""The arithmetic of the split-f16 GEMM path (insenticap_model_amd/csrc/gemm_f32.hip: gemm_h3_kernel), restated in numpy
so that its accuracy claim is checked without a GPU:

    x = hi + lo * 2^-11,  hi = f16(x),  lo = f16((x - hi) * 2^11)
    C = sum hi_a hi_b + 2^-11 * sum (hi_a lo_b + lo_a hi_b)          (lo_a lo_b dropped)

f16 x f16 products are exact in fp32/fp64, so the only errors are the representation error of the two planes and the
dropped term - both ~2^-24 relative.  Also pins the interleaved plane layout the kernels and producers agree on."""
import numpy as np

SCALE = 2048.0


def split(x):
    hi = x.astype(np.float16)
    lo = ((x - hi.astype(np.float32)) * np.float32(SCALE)).astype(np.float16)
    return hi, lo


def h3_matmul(a, w):
    ah, al = split(a)
    wh, wl = split(w)
    f = np.float64
    return ah.astype(f) @ wh.astype(f).T + (ah.astype(f) @ wl.astype(f).T + al.astype(f) @ wh.astype(f).T) / SCALE


def test_planes_represent_fp32_values_to_2_pow_minus_22():
    """|x| >= 2^-14 (f16's normal range for hi): relative error <= 2^-22 (two 11-bit significands); below it the
    representation degrades gracefully to an ABSOLUTE error <= 2^-35 (lo carries the residual scaled by 2^11)."""
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-16, 6, 200000))).astype(np.float32)
    x = x[np.abs(x) < 60000]
    hi, lo = split(x)
    back = hi.astype(np.float64) + lo.astype(np.float64) / SCALE
    err = np.abs(back - x.astype(np.float64))
    normal = np.abs(x) >= 2.0 ** -14
    assert normal.sum() > 50000 and (~normal).sum() > 10000
    assert (err[normal] / np.abs(x[normal].astype(np.float64))).max() <= 2.0 ** -22
    assert err[~normal].max() <= 2.0 ** -35


def test_three_product_contraction_is_fp32_accurate():
    rng = np.random.default_rng(1)
    M, N, K = 96, 80, 512
    a = rng.uniform(-1, 1, (M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    scale = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T
    h3 = h3_matmul(a, w)
    # fp32 FMA chain in k order (what v_mfma_f32_32x32x2_f32 computes), for comparison
    f32 = np.zeros((M, N), dtype=np.float32)
    for k in range(K):
        f32 = (f32.astype(np.float64) + a[:, k:k + 1].astype(np.float64) * w[None, :, k].astype(np.float64)).astype(np.float32)
    err_h3 = np.abs(h3 - ref)
    err_f32 = np.abs(f32.astype(np.float64) - ref)
    assert (err_h3 / scale).max() < 2.0 ** -21                        # representation + dropped lo*lo term
    assert err_h3.max() <= err_f32.max()                              # and below the fp32 chain's accumulated rounding
    assert np.sqrt((err_h3 ** 2).mean()) < np.sqrt((err_f32 ** 2).mean())


def plane_index(row, k, K):
    """include/insenticap_hip.h, isc_seg.A_hi: hi(row, k) at buf[row*2K + (k/32)*64 + k%32], lo = the same + 32."""
    return row * 2 * K + (k // 32) * 64 + (k % 32)


def test_interleaved_plane_layout_is_a_bijection_with_whole_lines_per_chunk():
    rows, K = 5, 96
    seen = set()
    for r in range(rows):
        for k in range(K):
            for off in (0, 32):
                i = plane_index(r, k, K) + off
                assert 0 <= i < rows * 2 * K and i not in seen
                seen.add(i)
    assert len(seen) == rows * 2 * K
    # the 32 hi and 32 lo halfs one 32-deep chunk consumes from a row are one contiguous 128-byte block
    for r in range(rows):
        for kb in range(K // 32):
            idx = sorted(plane_index(r, kb * 32 + j, K) + off for j in range(32) for off in (0, 32))
            assert idx == list(range(idx[0], idx[0] + 64)) and (idx[0] * 2) % 128 == 0
