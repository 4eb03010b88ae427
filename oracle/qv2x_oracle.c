/*
 * qv2x_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement, in plain C, of the integer / fixed-order arithmetic of QuantV2X's W8A8 hot path.
 * It is the checker the HIP kernels are compared with (bit-exact for every uint8 activation code
 * and every codebook index).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it; the product path (quantv2x_amd/) never does.
 *
 * What it restates (reference file:line):
 *   fake-quant  x_q = clamp(round(x/delta) + zp, 0, 255), x' = (x_q - zp) * delta
 *                                           opencood/quant/quant_layer.py:132-148 (UniformAffineQuantizer.forward)
 *   QuantModule out = act_quant(relu(conv(x', w') + b))          quant_layer.py:391-410
 *   QuantPFNLayer / QuantPillarVFE                                quant_block.py:589-715, pillar_vfe.py:105-155
 *   PointPillarScatter                                            point_pillar_scatter.py:19-75
 *   UMGMQuantizer.encode                                          codebook.py:106-131,231-239,330-337
 *
 * The reference evaluates these with fp32 torch kernels whose summation order is unspecified.  The
 * restatement fixes an order so that CPU and GPU agree bit for bit:
 *   - integer-valued products (u8 x u8 conv) are summed exactly in int32, then scaled:
 *       y = bias + sum_g float(T_g) * (dx_g * dw[co])         (float mul, float add; no fma)
 *   - fp32 dot products are ascending-k fmaf chains (what v_mfma_f32_32x32x2_f32 computes).
 * Against the reference this differs by fp32 re-association only (<= 1 LSB on rare elements; pinned by
 * tests/test_oracle_golden.py against vectors captured from the reference).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: the compiler must not fuse mul+add).
 */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

static inline float q_code(float y, float delta, float zp) {
    float t = rintf(y / delta) + zp;            /* round half to even, like torch.round */
    t = fmaxf(t, 0.0f);
    return fminf(t, 255.0f);
}

/* The output quantizer of the CONVOLUTION epilogues (3x3 / 1x1 / grouped / transposed convolutions, the end of a residual block) in its
 * deployed form (round 5):   code = clamp(rint(fma(y, fl(1 / delta), zp)), 0, 255)
 * -- UniformAffineQuantizer.forward (quant_layer.py:132-133: round(x / delta) + zero_point, clamp) with the division replaced by ONE fused
 * multiply-add with the fp32 reciprocal, which is what v_fma_f32 + v_cvt_pk_u8_f32 (round to nearest even, saturate) evaluate in two
 * instructions per output; the division-exact form cost the HIP epilogues 5.25 (the 64- and 128-channel layers are bound by
 * VALU issue).  The two forms give a different code only where p = x / delta lies close to a rounding boundary: within ~1.2e-7 |p| when
 * zp = 0 (every post-ReLU quantizer), and within ~ulp(p + zp) / 2 <= 3e-5 when zp != 0 -- there the division form rounds p to an integer
 * BEFORE zp is added, this form rounds p + zp once.  Measured on a 3x3 layer's 393 216 outputs
 * (tests/test_sanitizers_cpu.py::test_reciprocal_multiply_quantizer_against_the_division_form): 0 codes differ at zp = 0, 2 (5e-6) at
 * zp = 131, never by more than one LSB -- below the noise of the reference's own fp32 convolution sums and inside the golden-vector bound
 * (<= 1 LSB on < 5e-4 of the elements per layer, tests/test_oracle_golden.py).  Compile with -DORC_QDIV (make qdiv) for the division form.
 * The PFN's two quantizers and the heads keep q_code (pinned bit for bit on the reference's golden pillar codes). */
static inline float q_code_mul(float y, float delta, float zp) {
#ifdef ORC_QDIV
    return q_code(y, delta, zp);
#else
    const float rd = 1.0f / delta;
    float t = rintf(fmaf(y, rd, zp));
    t = fmaxf(t, 0.0f);
    return fminf(t, 255.0f);
#endif
}

/* ---------------------------------------------------------------------------------------------
 * a1: pillar feature net under QuantModel.  vf [M][P][4], coords [M][4] = (b, z, y, x), npts [M].
 * w [64][10] = dequantized (fake-quant) folded weights, b [64] folded BN bias.
 * Output: pillar_code [M][64] = code of the second activation quantizer after the max over P.
 * ------------------------------------------------------------------------------------------- */
ORC_API void orc_pfn(const float* vf, const int32_t* coords, const int32_t* npts, int M, int P,
                     const float* w, const float* b, float d1, float z1, float d2, float z2,
                     const float* vox, const float* off, uint8_t* pillar_code) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        const float* pts = vf + (size_t)m * P * 4;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int p = 0; p < P; ++p) { sx += pts[p * 4 + 0]; sy += pts[p * 4 + 1]; sz += pts[p * 4 + 2]; }
        const float n = (float)npts[m];
        const float mx = sx / n, my = sy / n, mz = sz / n;
        const float cx = (float)coords[m * 4 + 3] * vox[0] + off[0];
        const float cy = (float)coords[m * 4 + 2] * vox[1] + off[1];
        const float cz = (float)coords[m * 4 + 1] * vox[2] + off[2];
        float best[64];
        for (int c = 0; c < 64; ++c) best[c] = -1.0f;
        for (int p = 0; p < P; ++p) {
            const float mask = (npts[m] > p) ? 1.0f : 0.0f;
            const float x = pts[p * 4 + 0], y = pts[p * 4 + 1], z = pts[p * 4 + 2], it = pts[p * 4 + 3];
            float f[10];
            f[0] = x * mask; f[1] = y * mask; f[2] = z * mask; f[3] = it * mask;
            f[4] = (x - mx) * mask; f[5] = (y - my) * mask; f[6] = (z - mz) * mask;
            f[7] = (x - cx) * mask; f[8] = (y - cy) * mask; f[9] = (z - cz) * mask;
            for (int c = 0; c < 64; ++c) {
                float acc = 0.0f;
                for (int k = 0; k < 10; ++k) acc = fmaf(f[k], w[c * 10 + k], acc);
                float yv = acc + b[c];
                float q1 = q_code(yv, d1, z1);
                float y1 = (q1 - z1) * d1;
                y1 = fmaxf(y1, 0.0f);
                float q2 = q_code(y1, d2, z2);
                if (q2 > best[c]) best[c] = q2;
            }
        }
        for (int c = 0; c < 64; ++c) pillar_code[(size_t)m * 64 + c] = (uint8_t)best[c];
    }
}

/* a2: canvas [N][ny][nx][64] (pre-filled with the zero-point code) <- pillar codes at (b, y, x). */
ORC_API void orc_scatter(const uint8_t* pillar_code, const int32_t* coords, int M, int ny, int nx, uint8_t* canvas) {
    for (int m = 0; m < M; ++m) {
        const int32_t* c = coords + m * 4;
        size_t cell = ((size_t)c[0] * ny + c[2]) * nx + (size_t)(c[1] + c[3]);   /* z + y*nx + x with z == 0 */
        memcpy(canvas + cell * 64, pillar_code + (size_t)m * 64, 64);
    }
}

/* ---------------------------------------------------------------------------------------------
 * a3/a4/a5: 3x3 convolution on uint8 codes, zero padding 1 (in the dequantized domain: pad code = zx).
 * in  [N][H][W][Cin]  u8 codes; channel groups g = 0..G-1 cover [gc0[g], gc0[g]+gc[g]) with their own
 *     (dx_g implied by scale, zx_g).
 * wq  [Cout][Cin][3][3] u8 codes, zw [Cout]; scale [G][Cout] = dx_g * dw[co] (fp32 product made by the caller).
 * out [N][Ho][Wo][out_ct] u8 codes written at channel offset out_c0.
 * ------------------------------------------------------------------------------------------- */
ORC_API void orc_conv3x3(const uint8_t* in, int N, int H, int W, int Cin, int stride,
                         int G, const int32_t* gc0, const int32_t* gc, const int32_t* zx,
                         const uint8_t* wq, const int32_t* zw, int Cout, const float* scale, const float* bias,
                         int relu, float da, float za, uint8_t* out, int out_ct, int out_c0) {
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    /* weights repacked [Cout][3][3][Cin] so the inner loop is contiguous */
    int16_t* wr = (int16_t*)malloc((size_t)Cout * 9 * Cin * sizeof(int16_t));
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int t = 0; t < 9; ++t)
                wr[((size_t)co * 9 + t) * Cin + ci] = (int16_t)((int)wq[((size_t)co * Cin + ci) * 9 + t] - zw[co]);
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int yo = 0; yo < Ho; ++yo) {
            int16_t* xw = (int16_t*)malloc((size_t)9 * Cin * sizeof(int16_t));
            for (int xo = 0; xo < Wo; ++xo) {
                /* gather the window, already minus the group zero-point; padding contributes 0 */
                for (int t = 0; t < 9; ++t) {
                    const int yi = yo * stride + t / 3 - 1, xi = xo * stride + t % 3 - 1;
                    int16_t* dst = xw + (size_t)t * Cin;
                    if (yi < 0 || yi >= H || xi < 0 || xi >= W) { memset(dst, 0, (size_t)Cin * sizeof(int16_t)); continue; }
                    const uint8_t* src = in + (((size_t)n * H + yi) * W + xi) * Cin;
                    for (int g = 0; g < G; ++g)
                        for (int c = gc0[g]; c < gc0[g] + gc[g]; ++c) dst[c] = (int16_t)((int)src[c] - zx[g]);
                }
                uint8_t* o = out + (((size_t)n * Ho + yo) * Wo + xo) * out_ct + out_c0;
                for (int co = 0; co < Cout; ++co) {
                    const int16_t* wrow = wr + (size_t)co * 9 * Cin;
                    float y = bias[co];
                    for (int g = 0; g < G; ++g) {
                        int32_t T = 0;
                        for (int t = 0; t < 9; ++t) {
                            const int16_t* a = xw + (size_t)t * Cin + gc0[g];
                            const int16_t* bw = wrow + (size_t)t * Cin + gc0[g];
                            int32_t s = 0;
                            for (int c = 0; c < gc[g]; ++c) s += (int32_t)a[c] * (int32_t)bw[c];
                            T += s;
                        }
                        y = y + (float)T * scale[(size_t)g * Cout + co];
                    }
                    if (relu) y = fmaxf(y, 0.0f);
                    o[co] = (uint8_t)q_code_mul(y, da, za);
                }
            }
            free(xw);
        }
    free(wr);
}

/* ---------------------------------------------------------------------------------------------
 * a3 deblocks: ConvTranspose2d with kernel == stride == s.  The reference's weight quantizer scales
 * per dim 0 == C_in here (quant_layer.py:192-195 on a [Cin, Cout, s, s] weight), so the scale sits on
 * the reduction axis and the sum is evaluated in fp32:
 *     acc = 0; for ci ascending: acc = fmaf(float(xq[ci] - zx) * dx, wdeq[ci][co][i][j], acc); y = acc + bias[co]
 * in [N][H][W][Cin] u8; wdeq [Cin][Cout][s][s] f32 (dequantized); out [N][H*s][W*s][out_ct] at out_c0.
 * ------------------------------------------------------------------------------------------- */
ORC_API void orc_deconv(const uint8_t* in, int N, int H, int W, int Cin, float dx, int zx,
                        const float* wdeq, const float* bias, int Cout, int s, int relu, float da, float za,
                        uint8_t* out, int out_ct, int out_c0) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < H; ++y) {
            float* xf = (float*)malloc((size_t)Cin * sizeof(float));
            for (int x = 0; x < W; ++x) {
                const uint8_t* src = in + (((size_t)n * H + y) * W + x) * Cin;
                for (int c = 0; c < Cin; ++c) xf[c] = (float)((int)src[c] - zx) * dx;
                for (int i = 0; i < s; ++i)
                    for (int j = 0; j < s; ++j) {
                        uint8_t* o = out + (((size_t)n * H * s + (y * s + i)) * (W * s) + (x * s + j)) * out_ct + out_c0;
                        for (int co = 0; co < Cout; ++co) {
                            float acc = 0.0f;
                            for (int c = 0; c < Cin; ++c)
                                acc = fmaf(xf[c], wdeq[(((size_t)c * Cout + co) * s + i) * s + j], acc);
                            float yv = acc + bias[co];
                            if (relu) yv = fmaxf(yv, 0.0f);
                            o[co] = (uint8_t)q_code_mul(yv, da, za);
                        }
                    }
            }
            free(xf);
        }
}

/* ---------------------------------------------------------------------------------------------
 * fp32 linear on rows: out[r][j] = chain_k fmaf(in[r][k], w[j][k], acc), acc0 = b[j].   w [J][K]
 * ------------------------------------------------------------------------------------------- */
static void linear_rows(const float* in, int R, int K, const float* w, const float* b, int J, float* out) {
    /* loop order r, k, j with w transposed keeps every output's chain in ascending k and vectorizes over j */
    float* wt = (float*)malloc((size_t)K * J * sizeof(float));
    for (int j = 0; j < J; ++j) for (int k = 0; k < K; ++k) wt[(size_t)k * J + j] = w[(size_t)j * K + k];
#pragma omp parallel for schedule(static)
    for (int r = 0; r < R; ++r) {
        float* o = out + (size_t)r * J;
        for (int j = 0; j < J; ++j) o[j] = b ? b[j] : 0.0f;
        const float* x = in + (size_t)r * K;
        for (int k = 0; k < K; ++k) {
            const float xv = x[k];
            const float* wk = wt + (size_t)k * J;
            for (int j = 0; j < J; ++j) o[j] = fmaf(xv, wk[j], o[j]);
        }
    }
    free(wt);
}

ORC_API void orc_linear(const float* in, int R, int K, const float* w, const float* b, int J, float* out) {
    linear_rows(in, R, K, w, b, J, out);
}

/* sum of squares as four 64-wide ascending fmaf chains combined ((s0+s1)+(s2+s3)); D must be 256 */
static float sumsq256(const float* v) {
    float s[4];
    for (int q = 0; q < 4; ++q) { float a = 0.0f; for (int i = 0; i < 64; ++i) a = fmaf(v[q * 64 + i], v[q * 64 + i], a); s[q] = a; }
    return (s[0] + s[1]) + (s[2] + s[3]);
}

ORC_API void orc_codebook_c2(const float* codebook, int Kc, float* c2) {
    for (int k = 0; k < Kc; ++k) c2[k] = sumsq256(codebook + (size_t)k * 256);
}

/* |v|^2 over a segment of d dims in the op order of the HIP kernels: 64-wide ascending fma chains combined as a balanced tree
 * (d = 256: (p0 + p1) + (p2 + p3); d = 128: p0 + p1; d = 64: p0); a segment shorter than 64 is one chain of d. */
static float sumsq_seg(const float* v, int d) {
    if (d == 256) return sumsq256(v);
    if (d == 128) {
        float p[2];
        for (int q = 0; q < 2; ++q) { float a = 0.0f; for (int i = 0; i < 64; ++i) a = fmaf(v[q * 64 + i], v[q * 64 + i], a); p[q] = a; }
        return p[0] + p[1];
    }
    float a = 0.0f;
    for (int i = 0; i < d; ++i) a = fmaf(v[i], v[i], a);
    return a;
}

/* ---------------------------------------------------------------------------------------------
 * a6: residual multi-codebook encode of R rows of D = 256 | 64 floats, L levels, S = seg_num (m) segments of D / S dims,
 * Kc codes per segment and level (codebook.py:106-131: x.reshape(n, m, d), per-segment distance and argmin; :231-239, :330-337).
 * Per level: z = stage(x); q = qhead(z); per segment s: d_k = (|q_s|^2 + |C_s,k|^2) - 2 * (q_s . C_s,k); code_s = first argmin;
 *            x <- lhead(z) - concat_s C_s[code_s]   (not on the last level; codebook.py:192-201).
 * weights: per level pointers packed by the caller:  stage_w/b, qhead_w/b, lhead_w/b ([D][D] / [D]),
 *          codebook = the EXTENDED form [S * Kc][D]: row s * Kc + k holds C[s][k] in dims [s d, (s + 1) d) and zeros elsewhere
 *          (quantv2x_amd/ptq_state.py) -- q . row over all D dims is then the segment's own ascending fma chain, bit for bit
 *          (fmaf(q, 0, acc) = acc), which is how the HIP kernels evaluate it.  S = 1: the plain [Kc][D] codebook.
 * codes: [L * S][R] planes, plane l * S + s.
 * ------------------------------------------------------------------------------------------- */
ORC_API void orc_codebook_encode_seg(const float* x_in, int R, int L, int Kc, int D, int S,
                                     const float* const* stage_w, const float* const* stage_b,
                                     const float* const* qhead_w, const float* const* qhead_b,
                                     const float* const* lhead_w, const float* const* lhead_b,
                                     const float* const* codebook, uint8_t* codes /* [L * S][R] */,
                                     float* gap_out /* [L * S][R] top-2 gap or NULL */) {
    if ((D != 256 && D != 64) || S < 1 || D % S || Kc < 1 || Kc > 256) {
        /* (never a silent return: zero codes would look like an answer -- ADVICE r5) */
        fprintf(stderr, "orc_codebook_encode_seg: unsupported shape D %d, seg_num %d, dict_size %d\n", D, S, Kc);
        abort();
    }
    const int d = D / S, KE = S * Kc;
    float* x = (float*)malloc((size_t)R * D * sizeof(float));
    float* z = (float*)malloc((size_t)R * D * sizeof(float));
    float* q = (float*)malloc((size_t)R * D * sizeof(float));
    float* inter = (float*)malloc((size_t)R * KE * sizeof(float));
    float* c2 = (float*)malloc((size_t)KE * sizeof(float));
    memcpy(x, x_in, (size_t)R * D * sizeof(float));
    for (int l = 0; l < L; ++l) {
        linear_rows(x, R, D, stage_w[l], stage_b[l], D, z);
        linear_rows(z, R, D, qhead_w[l], qhead_b[l], D, q);
        linear_rows(q, R, D, codebook[l], NULL, KE, inter);
        for (int e = 0; e < KE; ++e) c2[e] = sumsq_seg(codebook[l] + (size_t)e * D + (size_t)(e / Kc) * d, d);
#pragma omp parallel for schedule(static)
        for (int r = 0; r < R; ++r) {
            for (int s = 0; s < S; ++s) {
                const float x2 = sumsq_seg(q + (size_t)r * D + (size_t)s * d, d);
                float best = INFINITY, second = INFINITY; int arg = 0;
                for (int k = 0; k < Kc; ++k) {
                    const float dist = (x2 + c2[s * Kc + k]) - 2.0f * inter[(size_t)r * KE + s * Kc + k];
                    if (dist < best) { second = best; best = dist; arg = k; }
                    else if (dist < second) second = dist;
                }
                codes[((size_t)l * S + s) * R + r] = (uint8_t)arg;
                if (gap_out) gap_out[((size_t)l * S + s) * R + r] = second - best;
            }
        }
        if (l < L - 1) {
            linear_rows(z, R, D, lhead_w[l], lhead_b[l], D, q);       /* q reused as latentHead(z) */
            const float* cb = codebook[l];
#pragma omp parallel for schedule(static)
            for (int r = 0; r < R; ++r)
                for (int s = 0; s < S; ++s) {
                    const float* c = cb + ((size_t)s * Kc + codes[((size_t)l * S + s) * R + r]) * D;
                    for (int j = s * d; j < (s + 1) * d; ++j) x[(size_t)r * D + j] = q[(size_t)r * D + j] - c[j];
                }
        }
    }
    free(x); free(z); free(q); free(inter); free(c2);
}

ORC_API void orc_codebook_encode_d(const float* x_in, int R, int L, int Kc, int D,
                                   const float* const* stage_w, const float* const* stage_b,
                                   const float* const* qhead_w, const float* const* qhead_b,
                                   const float* const* lhead_w, const float* const* lhead_b,
                                   const float* const* codebook, uint8_t* codes /* [L][R] */,
                                   float* gap_out /* [L][R] top-2 gap or NULL */) {
    orc_codebook_encode_seg(x_in, R, L, Kc, D, 1, stage_w, stage_b, qhead_w, qhead_b, lhead_w, lhead_b, codebook, codes, gap_out);
}

ORC_API void orc_codebook_encode(const float* x_in, int R, int L, int Kc,
                                 const float* const* stage_w, const float* const* stage_b,
                                 const float* const* qhead_w, const float* const* qhead_b,
                                 const float* const* lhead_w, const float* const* lhead_b,
                                 const float* const* codebook, uint8_t* codes, float* gap_out) {
    orc_codebook_encode_seg(x_in, R, L, Kc, 256, 1, stage_w, stage_b, qhead_w, qhead_b, lhead_w, lhead_b, codebook, codes, gap_out);
}

/* a7 as a table sum: out[r] = ((bias + T0[c0]) + T1[c1]) + T2[c2]; lut [L][Kc][256] */
ORC_API void orc_decode_lut(const uint8_t* codes, int R, int L, int Kc, const float* lut, const float* bias, float* out) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < R; ++r)
        for (int j = 0; j < 256; ++j) {
            float v = bias[j];
            for (int l = 0; l < L; ++l) v = v + lut[((size_t)l * Kc + codes[(size_t)l * R + r]) * 256 + j];
            out[(size_t)r * 256 + j] = v;
        }
}

ORC_API void orc_decode_lut_d(const uint8_t* codes, int R, int L, int Kc, int D, const float* lut, const float* bias, float* out) {
#pragma omp parallel for schedule(static)
    for (int r = 0; r < R; ++r)
        for (int j = 0; j < D; ++j) {
            float v = bias[j];
            for (int l = 0; l < L; ++l) v = v + lut[((size_t)l * Kc + codes[(size_t)l * R + r]) * D + j];
            out[(size_t)r * D + j] = v;
        }
}

/* a11: 1x1 heads on fp32 rows with fake-quant weights and (optionally) quantized output.
 * y = chain_k fmaf(x[k], w[co][k], acc), acc0 = bias[co]; out = (q(y) - za) * da   or y when !aq */
ORC_API void orc_heads(const float* x, int R, int K, const float* wdeq, const float* bias, int Cout,
                       int aq, float da, float za, float* out /* [R][Cout] */) {
    linear_rows(x, R, K, wdeq, bias, Cout, out);
    if (!aq) return;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < (size_t)R * Cout; ++i) out[i] = (q_code(out[i], da, za) - za) * da;
}

/* =============================================================================================
 * The UN-QUANTIZED model (fp32 mode of the product, quantv2x_amd/csrc/fp32_path.hip): the reference's plain fp32 forward
 * (pillar_vfe.py:105-155, base_bev_backbone.py:96-119, downsample_conv.py:26-51) with BatchNorm folded, every dot product ONE
 * fmaf chain in the order the f32 MFMA kernel walks K: groups of 8 consecutive k, inside a group k0, k4, k1, k5, k2, k6, k3, k7.
 * ============================================================================================= */
static const int ORC_K8[8] = {0, 4, 1, 5, 2, 6, 3, 7};

/* a1 + a2 in fp32: pillars -> [M][64] features (Linear 10 -> 64 with folded BN, ReLU, max over the P slots; zero-masked slots
 * contribute relu(bias)) */
ORC_API void orc_pfn_f32(const float* vf, const int32_t* coords, const int32_t* npts, int M, int P, const float* w, const float* b,
                         const float* vox, const float* off, float* out) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m) {
        const float* pts = vf + (size_t)m * P * 4;
        const int real = npts[m] < P ? npts[m] : P;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (int p = 0; p < real; ++p) { sx += pts[p * 4 + 0]; sy += pts[p * 4 + 1]; sz += pts[p * 4 + 2]; }
        const float n = (float)npts[m];
        const float mx = sx / n, my = sy / n, mz = sz / n;
        const float cx = (float)coords[m * 4 + 3] * vox[0] + off[0];
        const float cy = (float)coords[m * 4 + 2] * vox[1] + off[1];
        const float cz = (float)coords[m * 4 + 1] * vox[2] + off[2];
        for (int c = 0; c < 64; ++c) {
            float best = -INFINITY;
            for (int p = 0; p < real; ++p) {
                const float x = pts[p * 4 + 0], y = pts[p * 4 + 1], z = pts[p * 4 + 2], it = pts[p * 4 + 3];
                const float f[10] = {x, y, z, it, x - mx, y - my, z - mz, x - cx, y - cy, z - cz};
                float acc = 0.0f;
                for (int k = 0; k < 10; ++k) acc = fmaf(f[k], w[c * 10 + k], acc);
                best = fmaxf(best, acc + b[c]);
            }
            if (real < P) best = fmaxf(best, b[c]);
            out[(size_t)m * 64 + c] = fmaxf(best, 0.0f);
        }
    }
}

/* 3x3 convolution, zero padding 1, stride 1 | 2 (deconv == 0), or ConvTranspose2d with kernel == stride == s (deconv == 1).
 * in [N][H][W][cin_total] (window [cin0, cin0 + cin)), w [cols][K] row-major with K = 9 * cin (k = tap * cin + ci) or cin,
 * cols = cout or (i*s + j)*cout + co; out [N][Ho][Wo][out_ct] at channel offset out_c0. */
ORC_API void orc_gemm_f32(const float* in, int N, int H, int W, int cin_total, int cin0, int cin, int stride, int cout, int deconv,
                          const float* w, const float* bias, int relu, float* out, int out_ct, int out_c0) {
    const int taps = deconv ? 1 : 9, K = taps * cin;
    const int Ho = deconv ? H * stride : (H + 2 - 3) / stride + 1, Wo = deconv ? W * stride : (W + 2 - 3) / stride + 1;
    const int rows = deconv ? N * H * W : N * Ho * Wo;
#pragma omp parallel
    {
        float* xk = (float*)malloc((size_t)K * sizeof(float));
#pragma omp for schedule(static)
        for (int m = 0; m < rows; ++m) {
            const int per = deconv ? H * W : Ho * Wo, rw = deconv ? W : Wo;
            const int img = m / per, rem = m % per, ry = rem / rw, rx = rem % rw;
            for (int t = 0; t < taps; ++t) {
                const int yy = deconv ? ry : ry * stride + t / 3 - 1, xx = deconv ? rx : rx * stride + t % 3 - 1;
                const int inside = yy >= 0 && yy < H && xx >= 0 && xx < W;
                const float* px = in + ((size_t)(img * H + (inside ? yy : 0)) * W + (inside ? xx : 0)) * cin_total + cin0;
                for (int c = 0; c < cin; ++c) xk[t * cin + c] = inside ? px[c] : 0.0f;
            }
            const int ncols = deconv ? stride * stride * cout : cout;
            for (int col = 0; col < ncols; ++col) {
                const float* wr = w + (size_t)col * K;
                float acc = 0.0f;
                for (int g = 0; g < K; g += 8)
                    for (int e = 0; e < 8; ++e) acc = fmaf(xk[g + ORC_K8[e]], wr[g + ORC_K8[e]], acc);
                int co = col, oy = ry, ox = rx;
                if (deconv) { const int ij = col / cout; co = col % cout; oy = ry * stride + ij / stride; ox = rx * stride + ij % stride; }
                float y = acc + bias[co];
                if (relu) y = fmaxf(y, 0.0f);
                out[((size_t)(img * Ho + oy) * Wo + ox) * out_ct + out_c0 + co] = y;
            }
        }
        free(xk);
    }
}

/* =============================================================================================
 * f3: the layers the HEAL Pyramid model adds (quant_block.py:68-131 QuantBasicBlock / QuantBottleneck, :462-549 QuantPyramidFusion).
 * ============================================================================================= */

/* k x k convolution (k = 1 | 3, zero padding k / 2, stride 1 | 2, `groups` channel groups as torch's Conv2d) on uint8 codes with
 * ONE input quantizer (dx implied by scale, zx):   T = sum (x - zx) * (w - zw[co])  exact;   y = bias[co] + float(T) * scale[co].
 *   mode 0: out_u8 = q_code(relu ? max(y, 0) : y)      (QuantModule with its output quantizer, quant_layer.py:391-410)
 *   mode 1: out_f32 = y                                (disable_act_quant: the end of a residual branch, the 1x1 shortcut)
 * in [N][H][W][Cin]; wq [Cout][Cin/groups][k][k] u8 codes; outputs [N][Ho][Wo][Cout]. */
ORC_API void orc_convg(const uint8_t* in, int N, int H, int W, int Cin, int zx, int k, int stride, int groups,
                       const uint8_t* wq, const int32_t* zw, int Cout, const float* scale, const float* bias,
                       int mode, int relu, float da, float za, uint8_t* out_u8, float* out_f32) {
    const int pad = k / 2, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    const int cg = Cin / groups, og = Cout / groups;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int yo = 0; yo < Ho; ++yo)
            for (int xo = 0; xo < Wo; ++xo)
                for (int co = 0; co < Cout; ++co) {
                    const int g = co / og;
                    int32_t T = 0;
                    for (int i = 0; i < k; ++i)
                        for (int j = 0; j < k; ++j) {
                            const int yi = yo * stride + i - pad, xi = xo * stride + j - pad;
                            if (yi < 0 || yi >= H || xi < 0 || xi >= W) continue;       /* zero padding: (zx - zx) * w */
                            const uint8_t* src = in + (((size_t)n * H + yi) * W + xi) * Cin + g * cg;
                            const uint8_t* wr = wq + (((size_t)co * cg) * k + i) * k + j;
                            for (int c = 0; c < cg; ++c) T += ((int)src[c] - zx) * ((int)wr[(size_t)c * k * k] - zw[co]);
                        }
                    float y = bias[co] + (float)T * scale[co];
                    const size_t o = (((size_t)n * Ho + yo) * Wo + xo) * Cout + co;
                    if (mode == 1) { out_f32[o] = y; continue; }
                    if (relu) y = fmaxf(y, 0.0f);
                    out_u8[o] = (uint8_t)q_code_mul(y, da, za);
                }
}

/* ConvTranspose2d(kernel == stride == s) or (s == 1) a 1x1 convolution on fp32 rows with fake-quantized weights:
 *   acc = 0; ci ascending: acc = fmaf(x[ci], wdeq[ci][co][i][j], acc);  y = acc + bias[co];  out = q_code(relu(y))
 * xf [N*H*W][Cin] f32 (the decoded feature, or a fused map -- neither sits on a quantizer grid); out as orc_deconv. */
ORC_API void orc_deconv_f32in(const float* xf, int N, int H, int W, int Cin, const float* wdeq, const float* bias, int Cout, int s,
                              int relu, float da, float za, uint8_t* out, int out_ct, int out_c0) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const float* src = xf + (((size_t)n * H + y) * W + x) * Cin;
                for (int i = 0; i < s; ++i)
                    for (int j = 0; j < s; ++j) {
                        uint8_t* o = out + (((size_t)n * H * s + (y * s + i)) * (W * s) + (x * s + j)) * out_ct + out_c0;
                        for (int co = 0; co < Cout; ++co) {
                            float acc = 0.0f;
                            for (int c = 0; c < Cin; ++c) acc = fmaf(src[c], wdeq[(((size_t)c * Cout + co) * s + i) * s + j], acc);
                            float yv = acc + bias[co];
                            if (relu) yv = fmaxf(yv, 0.0f);
                            o[co] = (uint8_t)q_code_mul(yv, da, za);
                        }
                    }
            }
}

/* the end of a residual block: code = q_code(max(y + res, 0))   (out += residual; ReLU; act_quantizer) */
ORC_API void orc_add_relu_quant(const float* y, const float* res, size_t n, float da, float za, uint8_t* out) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) out[i] = (uint8_t)q_code_mul(fmaxf(y[i] + res[i], 0.0f), da, za);
}
