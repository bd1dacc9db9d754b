// render_backward_v3.h -- generation 3 of the backward composite (round 1's product kernel: one cross-lane packed reduction and one
// atomic instruction per Gaussian visit; 185 us at workload B against generation 5's 112 us).  Superseded; compiled into the
// library only by `make EXPERIMENTS=1` (included by render.hip, whose staging helpers it uses) and selected with
// splat_debug_option(3, 3) for A/B timing (scripts/ab_k7_gen.py).
// NE: list entries per loop trip.  With NE = 2 the 2 x NV partial sums of two consecutive entries are reduced together
// (2 x 10 sums = five full packed groups instead of 2 x 3 padded ones; 2 x 7 = four instead of 2 x 2), leave the wave in
// ONE atomic instruction (two accumulator lines) and share the scalar loop overhead.
// OPAC: the caller needs S6 = sum G dL/dalpha (dL/dopacity); the fused tracking iteration does not (opacities get no update
// while the camera is tracked), and 2 x (5 + 1) sums are three full packed groups with two entries per trip.
// BG: the background colour can be non-zero (the fused iteration renders on a zero background and drops that term).
template <int C, int CS, unsigned DMASK, unsigned SMASK, int NE, bool OPAC = true, bool BG = true>
__global__ __launch_bounds__(256) void render_backward_kernel(SplatCamera cam, const float *colors, SplatState st,
                                                              const float *dL_dcolor, float *accum, int T, int per_xcd) {
    constexpr int FP = (C + 3) / 4 * 4;
    constexpr int NS = popcount_c(SMASK);
    constexpr int NB = OPAC ? 6 : 5;          // geometric sums: S1..S5 (+ S6)
    constexpr int NV = NB + NS;               // partial sums per Gaussian
    constexpr int NVT = NE * NV;              // partial sums per loop trip
    constexpr int NG = (NVT + 3) / 4;         // packed reduction groups per loop trip
    __shared__ Batch<FP> B;
    __shared__ unsigned s_wmax[4];
    const int tile = block_tile(per_xcd, T);
    if (tile < 0) return;
    const int W = cam.image_width, H = cam.image_height;
    const int gx = (W + kTile - 1) / kTile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tx = tile % gx, ty = tile / gx;
    const int px = tx * kTile + (wave & 1) * 8 + (lane & 7), py = ty * kTile + (wave >> 1) * 8 + (lane >> 3);
    const float fpx = (float)px, fpy = (float)py;
    const float tile_x0 = (float)(tx * kTile), tile_y0 = (float)(ty * kTile);
    const bool inside = px < W && py < H;
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)py * W + px;

    // Per pixel: running transmittance, and the scalar form of the "colour behind" recursion:
    // with cdot_i = sum_ch c_i[ch] dL/dC[ch], the reference's accum_rec[ch] only ever enters through
    // behind = sum_ch accum_rec[ch] dL/dC[ch], which obeys behind <- a_prev cdot_prev + (1-a_prev) behind.
    const float Tfin = inside ? st.final_T[pix] : 0.f;
    float Tr = Tfin;
    const unsigned last = inside ? (unsigned)st.n_contrib[pix] : 0u;
    float dpix[C], bgdot = 0.f, behind = 0.f, lcdot = 0.f, lalpha = 0.f;
    bool has_bg = false;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
        dpix[ch] = 0.f;
        if ((DMASK >> ch) & 1u) {
            dpix[ch] = inside ? dL_dcolor[ch * HW + pix] : 0.f;
            if constexpr (BG) {
                has_bg |= cam.bg[ch] != 0.f;
                bgdot += cam.bg[ch] * dpix[ch];
            }
        }
    }
    const unsigned wmax = (unsigned)__builtin_amdgcn_readfirstlane((int)wave_max_u32(last));   // deepest contributor of this quadrant
    if (lane == 0) s_wmax[wave] = wmax;
    __syncthreads();
    const unsigned tmax = (unsigned)__builtin_amdgcn_readfirstlane((int)max(max(s_wmax[0], s_wmax[1]), max(s_wmax[2], s_wmax[3])));
    if (tmax == 0) return;                                     // uniform over the workgroup
    const unsigned lo = st.tile_stride > 0 ? (unsigned)tile * (unsigned)st.tile_stride : st.tile_base[tile];
    const int nb = (int)((tmax + kBatch - 1) / kBatch);
    // Publish: after the packed reductions every lane of row r holds value row_value(r) of each group.  Lane 16 r + g takes
    // group g's value, so that ONE global_atomic_add_f32 carries all 6 + |SMASK| sums of the Gaussian to its 64-byte
    // accumulator line.  The L2 atomic units retire ~21 line-requests per ns however many floats of the line a request
    // carries (scripts/micro/atomic_bench.hip): one instruction per visit instead of one per group cuts the kernel's
    // 2.9 M line-requests per launch to 0.97 M.
    const int pub_g = lane & 15;
    int doff = -1;                            // accumulator slot this lane publishes
    int pub_e = 0;                            // ... of which entry of the trip
#pragma unroll
    for (int grp = 0; grp < NG; ++grp)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kt = 4 * grp + row_value(r);          // index into the trip's NVT sums
            const int ent = kt / NV, k = kt - ent * NV;
            const int slot = kt >= NVT ? -1 : (k < NB ? k : 6 + nth_set_bit(SMASK, k - NB < 0 ? 0 : k - NB));
            if (pub_g == grp && (lane >> 4) == r) { doff = slot; pub_e = ent; }
        }
    unsigned long long pub_m[NE];             // publishing lanes of each entry
#pragma unroll
    for (int t = 0; t < NE; ++t) pub_m[t] = __builtin_amdgcn_ballot_w64(doff >= 0 && pub_e == t);

    Staged<FP> pre;
    {
        const unsigned e = (unsigned)((nb - 1) * kBatch + tid);
        gather<C, CS, false, FP>(pre, st, colors, lo + e, e < tmax, tile_x0, tile_y0);
    }
    for (int bi = nb - 1; bi >= 0; --bi) {
        if (bi < nb - 1) __syncthreads();           // every wave has finished reading the previous batch
        commit(B, pre, tid, 0u);
        __syncthreads();
        const bool more = bi > 0;
        if (more) gather<C, CS, false, FP>(pre, st, colors, lo + (unsigned)((bi - 1) * kBatch + tid), true, tile_x0, tile_y0);
        const int base = bi * kBatch;
        const int lim = (int)wmax - base;                      // entries [0, lim) of this batch can matter to this wave
#pragma unroll 1
        for (int w = 3; w >= 0; --w) {
            const int k = lim - 64 * w;
            if (k <= 0) continue;
            unsigned long long bits = mask_word(B, wave, w);
            if (k < 64) bits &= (1ull << k) - 1ull;
            while (bits != 0) {
                float s[NG * 4];
                unsigned ids[NE];
                unsigned long long any_m = 0, exec_m = 0;
#pragma unroll
                for (int v = 0; v < NG * 4; ++v) s[v] = 0.f;
#pragma unroll
                for (int t = 0; t < NE; ++t) {
                    ids[t] = 0;
                    if (bits == 0) continue;                    // odd number of entries: the second half of the trip is empty
                    const int j = 63 - __builtin_clzll(bits);
                    bits &= ~(1ull << j);
                    const int e = w * 64 + j;
                    Entry<FP> cur;
                    read_entry(B, e, cur);
                    ids[t] = cur.id;
                    const unsigned pos = (unsigned)(base + e + 1);
                    const float dx = cur.mux - fpx, dy = cur.muy - fpy;
                    const float p2 = dx * (cur.ga.x * dx + cur.ga.y * dy) + cur.ga.z * dy * dy;
                    const float G = fast_exp2(p2);
                    const float alpha = fminf(kAlphaMax, cur.ga.w * G);
                    const unsigned long long live_m = __builtin_amdgcn_ballot_w64(pos <= last) & __builtin_amdgcn_ballot_w64(p2 <= 0.f) &
                                                      __builtin_amdgcn_ballot_w64(alpha >= kAlphaMin);
                    if (live_m == 0) continue;
                    any_m |= live_m;
                    exec_m |= pub_m[t];
                    const bool live = lane_of(live_m);
                    const float rcp = __builtin_amdgcn_rcpf(1.f - alpha);
                    const float Tn = Tr * rcp;                     // transmittance in front of this Gaussian
                    // sum_ch colour[ch] * dL/dC[ch], summed pairwise (packed multiplies / adds)
                    float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ch = 0; ch < C; ++ch)
                        if ((DMASK >> ch) & 1u) part[ch & 3] += cur.feat[ch] * dpix[ch];
                    const float cdot = (part[0] + part[2]) + (part[1] + part[3]);
                    const float bh = lalpha * lcdot + (1.f - lalpha) * behind;
                    float dL_dalpha = (cdot - bh) * Tn;
                    if constexpr (BG)
                        if (has_bg) dL_dalpha += (-Tfin * rcp) * bgdot;
                    // selects (not multiplies by 0) so that a non-live lane can never inject inf * 0
                    const float Gl = live ? G : 0.f;
                    const float wgt = live ? alpha * Tn : 0.f;
                    const float q = live ? cur.ga.w * dL_dalpha : 0.f;   // dL/dG
                    const float gdx = Gl * dx, gdy = Gl * dy;
                    const float qgx = q * gdx, qgy = q * gdy;
                    float *sv = s + t * NV;
                    sv[0] = qgx;
                    sv[1] = qgy;
                    sv[2] = qgx * dx;
                    sv[3] = qgx * dy;
                    sv[4] = qgy * dy;
                    if constexpr (OPAC) sv[5] = Gl * dL_dalpha;
#pragma unroll
                    for (int n = 0; n < NS; ++n) sv[NB + n] = wgt * dpix[nth_set_bit(SMASK, n)];
                    Tr = live ? Tn : Tr;
                    behind = live ? bh : behind;
                    lcdot = live ? cdot : lcdot;
                    lalpha = live ? alpha : lalpha;
                }
                if (any_m != 0) {
                    // all packed reductions first (independent chains interleave), then one publish
                    // (timing ablation, round 1: without the atomics -11..-23 us, without the cross-lane reduction -40..-60 us,
                    //  without both -100..-120 us of ~245 us per launch)
                    float r[NG];
#pragma unroll
                    for (int grp = 0; grp < NG; ++grp)
                        r[grp] = wave_reduce4_packed(s[4 * grp], s[4 * grp + 1], s[4 * grp + 2], s[4 * grp + 3]);
                    float pv = r[0];
#pragma unroll
                    for (int grp = 1; grp < NG; ++grp) pv = pub_g == grp ? r[grp] : pv;
                    unsigned id = ids[0];
                    if constexpr (NE > 1) id = pub_e == 1 ? ids[1] : ids[0];
                    if (lane_of(exec_m)) atomicAdd(accum + (size_t)id * SPLAT_GRAD_STRIDE + doff, pv);
                }
            }
        }
    }
}

