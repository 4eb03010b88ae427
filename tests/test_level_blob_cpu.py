"""The level blob of the encode kernels (include/qv2x.h, codebook_encode.h) on the host side: both sections hold the SAME four matrices, each in
the order its kernel form reads -- the workgroup form's [K/4][cols][k0, k2, k1, k3] and the wave form's
[pair][group][tile][lane = 32 h + c][step] = W[64 pair + 32 tile + c][8 group + 2 step + h] (codebook_encode_wave.hip)."""
import numpy as np

from quantv2x_amd.engine import ENC_WAVE_PAD, _pack_k4p, _pack_wave, _pack_wave_seg, wave_section


def test_wave_order_is_the_mfma_a_operand_order():
    rng = np.random.default_rng(0)
    w = rng.standard_normal((256, 256)).astype(np.float32)
    p = _pack_wave(w)
    assert p.shape == (4, 32, 2, 2, 32, 4)                         # [pair][group][tile][h][c][step]
    for (P, g, t, h, c, s) in [(0, 0, 0, 0, 0, 0), (3, 31, 1, 1, 31, 3), (1, 7, 0, 1, 5, 2), (2, 19, 1, 0, 30, 1)]:
        assert p[P, g, t, h, c, s] == w[64 * P + 32 * t + c, 8 * g + 2 * s + h]
    # one group of one tile = 64 lanes x 16 bytes = 1 KiB contiguous; a pair's 32 groups x 2 tiles = 64 KiB: the stream the kernel walks
    assert p[0, 0, 0].size * 4 == 1024 and p[0].size * 4 == 64 * 1024
    # every element exactly once
    assert np.array_equal(np.sort(p.reshape(-1)), np.sort(w.reshape(-1)))


def test_a_dictionary_of_96_codes_is_padded_to_whole_tile_pairs():
    rng = np.random.default_rng(1)
    cb = rng.standard_normal((96, 256)).astype(np.float32)
    p = _pack_wave(cb)
    assert p.shape[0] == 2                                           # 128 rows: the second tile of the last pair is zeros
    assert not p[1, :, 1].any() and p[1, :, 0].any()
    assert p[1, 3, 0, 1, 7, 2] == cb[64 + 7, 8 * 3 + 2 * 2 + 1]


def test_both_sections_hold_the_same_matrices_and_the_size_the_library_states():
    from quantv2x_amd import lib as L
    lib = L.load()
    rng = np.random.default_rng(2)
    for kc in (32, 64, 96, 128):
        mats = [rng.standard_normal((256, 256)).astype(np.float32) for _ in range(3)]
        cb = rng.standard_normal((kc, 256)).astype(np.float32)
        wg = sum(_pack_k4p(m).size + 256 for m in mats) + _pack_k4p(cb).size + cb.size + kc
        sec = wave_section(mats[0], mats[1], mats[2], cb)
        assert wg + sec.size == lib.qv2x_codebook_level_floats(kc)
        assert sec.size == 3 * 65536 + (kc + 63) // 64 * 64 * 256 + ENC_WAVE_PAD and not sec[-ENC_WAVE_PAD:].any()
        # stage | qhead | codebook | lhead, in the order the kernel consumes them
        st = sec[:65536].reshape(4, 32, 2, 2, 32, 4)
        k4 = _pack_k4p(mats[0])                                       # [K/4][cols][k0, k2, k1, k3]
        for (col, k) in [(0, 0), (255, 255), (100, 37), (33, 130)]:
            a = st[col // 64, k // 8, (col % 64) // 32, k % 2, col % 32, (k % 8) // 2]
            b = k4[k // 4, col, [0, 2, 1, 3].index(k % 4)]
            assert a == b == mats[0][col, k]
        lh = sec[2 * 65536 + (kc + 63) // 64 * 64 * 256:][:65536].reshape(4, 32, 2, 2, 32, 4)
        assert lh[1, 2, 1, 0, 3, 1] == mats[2][64 + 32 + 3, 8 * 2 + 2 * 1 + 0]


def test_segmented_codebooks_stream_their_diagonal_blocks_only():
    """seg_num m > 1 (codebook_encode_wave.hip, SEGS > 1): the wave form's codebook stream = the m diagonal [kc x 256 / m] blocks of the extended
    codebook, a tile pair = code tile P of two segments; the section keeps the size the library states, lhead follows the shorter stream"""
    from quantv2x_amd import lib as L
    from quantv2x_amd.ptq_state import extended_codebook
    lib = L.load()
    rng = np.random.default_rng(3)
    for segs, kc in ((2, 256), (4, 64), (2, 64), (4, 128)):
        d = 256 // segs
        cb = extended_codebook(rng.standard_normal((segs, kc, d)).astype(np.float32))
        assert cb.shape == (segs * kc, 256)
        p = _pack_wave_seg(cb, segs)
        assert p.shape == (segs // 2, kc // 32, 32 // segs, 2, 2, 32, 4)           # [segment pair][tile][group][segment of the pair][h][c][step]
        for (pr, P, g, t, h, c, st) in [(0, 0, 0, 0, 0, 0, 0), (segs // 2 - 1, kc // 32 - 1, 32 // segs - 1, 1, 1, 31, 3), (0, 1, 3, 1, 0, 5, 2)]:
            seg = 2 * pr + t
            assert p[pr, P, g, t, h, c, st] == cb[seg * kc + 32 * P + c, d * seg + 8 * g + 2 * st + h]
        assert p.size * segs == cb.size                                           # 1 / m of the dense stream: the zero blocks are never read
        mats = [rng.standard_normal((256, 256)).astype(np.float32) for _ in range(3)]
        sec = wave_section(mats[0], mats[1], mats[2], cb, segs)
        wg = sum(_pack_k4p(m).size + 256 for m in mats) + _pack_k4p(cb).size + cb.size + segs * kc
        assert wg + sec.size == lib.qv2x_codebook_level_floats(segs * kc)
        lh = sec[2 * 65536 + p.size:][:65536].reshape(4, 32, 2, 2, 32, 4)
        assert lh[1, 2, 1, 0, 3, 1] == mats[2][64 + 32 + 3, 8 * 2 + 2 * 1 + 0]
        assert not sec[3 * 65536 + p.size:].any()
