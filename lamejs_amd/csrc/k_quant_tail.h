// k_quant_tail.h -- how a launch of the persistent quantization kernel ends (g_quant, lhip_api.cpp).
//
// What it is for: a launch of the persistent quantization kernel ends with ~2 ms in which waves leave one by one (a two-channel frame is
// 1.5-2 ms of one wave; measured: 1.29 ms of idle per wave and launch, profiles/r03_quant_wave_tail.txt).  Here a wave that finds the frame
// dispenser empty stays and HELPS the waves of its workgroup that are still inside a frame: every owner offers the second channel of the
// granule it is starting (the channels of a granule are independent given the granule's bit budget -- the reference's loop order,
// Quantize.js:1406-1466, does channel 0 then channel 1 with nothing passing between them but the reservoir count afterwards) in a per-
// workgroup table in LDS; a helper claims an offer, quantizes that granule-channel on its own LDS record and hands back the bits it used,
// the next bin-search seed and the block type; an offer nobody has claimed when the owner has finished channel 0 is withdrawn and the owner
// does channel 1 itself -- the worst case is the serial order of kb_quant.  Which wave quantizes a granule-channel is not observable in the
// output: the unit is a pure function of (frame, granule, channel, seed, target bits) and the global records it writes are the same.
//
// State of an offer: 0 none, 1 open, 2 claimed, 3 done.  Owner: fields, release, 1 ... CAS(1 -> 0) withdraws; else wait for 3, read, 0.
// Helper: CAS(1 -> 2), acquire, unit, results, release, 3.  Helpers leave when no wave of the workgroup can draw a frame any more.
// Tested in the 64-lane simulation as a real 8-wave workgroup (tests/hostsim: liblamejs_wavesim.so runs two-channel batches this way).  Measured
// (profiles/r03_tail_help_experiment_first_device_runs.txt, profiles/r04_ab_tail_help.txt): idle at the end of a launch 1.29 -> 0.81 ms per wave,
// step 45.6 -> 45.2 ms, a 16384-frame chunk of the host path -2.3 %; shipped in round 4 after the full sweep.
#pragma once
namespace lhip {

struct TailOffer { int state, fslot, gr, mode_ext, targ, seed_start, seed_step, gr0_bt, bits, next_start, next_step, block_type, active, pad_[3]; };
struct TailShare { int drawing, pad_[3]; TailOffer offer[8]; };

#if defined(LHIP_WAVESIM)
// how often each way was taken (printed at exit with LAMEJS_TAILHELP_STATS=1: the simulation must exercise both)
struct TailStats { long claimed = 0, withdrawn = 0; ~TailStats() { if (getenv("LAMEJS_TAILHELP_STATS")) fprintf(stderr, "tail-help: %ld granule-channels by helpers, %ld offers withdrawn\n", claimed, withdrawn); } };
inline TailStats& tail_stats() { static TailStats t; return t; }
#define LHIP_TAIL_COUNT(f) do { if (lane == 0) tail_stats().f++; } while (0)
#else
#define LHIP_TAIL_COUNT(f) do { } while (0)
#endif
// (the workgroup-scope signalling primitives wg_cas / wg_load / wg_store / wg_idle / wg_acquire and LHIP_SPIN_GUARD live in lhip_wave.h)

// kb_quant for the persistent kernel's speculative pass (chain == 0, no reservoir) with the second channel of every granule on offer
LHIP_DEV void kb_quant_th(const Tables& T, const PowBase& pb10, const Workspace& W, const StreamDesc* SD, int fslot, int lane, QuantLds& L,
                          const QuantTabs& Q, int* hint, TailShare& TS, int wv) {
    lane = lane_anew(lane);
    const int C = T.channels_out;
    const int st = uni(W.fslot_stream[fslot]);            // (frame-level values in scalar registers: as vector values they were parked in scratch memory across the search)
    const StreamDesc sd = SD[st];
    const int k = uni(fslot - sd.fslot0 - 1);
    if (k < 0) return;
    const int fidx = uni(sd.out_slot0 + k);
#ifdef LHIP_PHASE_PROF
    const unsigned long long ph_total0_ = __builtin_amdgcn_s_memtime();     // L.prof is zeroed / flushed once per wave by g_quant
#endif
    const double ath_adjust = unid(W.ath_adjust[fslot]);      // (scalar registers: as a vector value it was parked in scratch memory across the search)
    const int padding = uni(frame_padding(T, sd, k));
    const int mean_bits = uni((frame_bits_of(T, padding) - T.sideinfo_len * 8) / T.mode_gr);
    Seed seed0, seed1;
    seed0.start = seed1.start = W.spec_start; seed0.step = seed1.step = W.spec_step;
    if (k == 0) {
        seed0 = seed_before(W, sd, C, k, 0, 0);
        if (C > 1) seed1 = seed_before(W, sd, C, k, 0, 1);
        seed0.start = uni(seed0.start); seed0.step = uni(seed0.step); seed1.start = uni(seed1.start); seed1.step = uni(seed1.step);
    } else if (hint[0] == st) {
        if (hint[1] >= 0) { seed0.start = hint[1]; seed0.step = 2; }
        if (C > 1 && hint[2] >= 0) { seed1.start = hint[2]; seed1.step = 2; }
    }
    int gr0_bt0 = 0, gr0_bt1 = 0;
    const int Cp = T.psy_channels;
    const int mode_ext = (T.mode == 1) ? q_ms_decision(T, W, sd, k, lane, L) : 0;
    int ResvSize = 0;
    q_ath_pseudo(T, pb10, ath_adjust, lane, L, Q);
    TailOffer& o = TS.offer[wv];
    for (int gr = 0; gr < T.mode_gr; gr++) {
        const int gslot = sd.gslot0 + 1 + T.mode_gr * k + gr;
        int targ[2] = {0, 0};
        const double max_bits = (double)targ_bits_for(T, mean_bits, gr, ResvSize, targ);
        if (mode_ext == 2) {
            const float* te = W.tot_ener + (int64_t)(gslot - 1) * 4;
            double r = (double)te[2] + (double)te[3];
            if (r > 0) r = (double)te[3] / r;
            q_reduce_side(targ, r, mean_bits, max_bits);
        }
        const int targ0 = uni(targ[0]), targ1 = uni(targ[1]);
        if (C == 2) {                                  // channel 1 of this granule: on offer while this wave does channel 0
            if (lane == 0) { o.fslot = fslot; o.gr = gr; o.mode_ext = mode_ext; o.targ = targ1; o.seed_start = seed1.start; o.seed_step = seed1.step; o.gr0_bt = gr0_bt1; }
            wg_store(&o.state, 1, lane);
        }
        for (int ch = 0; ch < C; ch++) {               // ONE call site of the unit (code size decides the instruction cache's behaviour here)
            if (ch == 1 && wg_cas(&o.state, 1, 0, lane) != 1) {
                // a helper has channel 1: wait for its results (it wrote this channel's gr0 scalefactors into L.sf_gr0[1])
                { long spins = 0; while (wg_load(&o.state, lane) != 3) { wg_idle(); LHIP_SPIN_GUARD(spins); } }
                wg_acquire();
                const int bits = uni(o.bits), act = uni(o.active), ns = uni(o.next_start), nstep = uni(o.next_step), bt = uni(o.block_type);
                if (act) { seed1.start = ns; seed1.step = nstep; }
                ResvSize = uni(ResvSize - bits);
                if (gr == 0) gr0_bt1 = bt;
                wg_store(&o.state, 0, lane);
                continue;
            }
            if (ch == 1) LHIP_TAIL_COUNT(withdrawn);   // nobody took it: the serial order
            const UnitOut u = q_unit(T, pb10, W, C, Cp, fidx, gslot, gr, ch, mode_ext, ath_adjust, ch == 0 ? targ0 : targ1, ch == 0 ? seed0 : seed1,
                                     ch == 0 ? gr0_bt0 : gr0_bt1, lane, L, Q);
            if (u.active) { if (ch == 0) seed0 = u.next; else seed1 = u.next; }
            ResvSize = uni(ResvSize - u.bits);
            if (gr == 0) { if (ch == 0) gr0_bt0 = u.block_type; else gr0_bt1 = u.block_type; }
        }
    }
    hint[0] = st; hint[1] = seed0.start; hint[2] = C > 1 ? seed1.start : -1;
#ifdef LHIP_PHASE_PROF
    if (lane == 0) { L.prof[PH_TOTAL] += (unsigned int)(__builtin_amdgcn_s_memtime() - ph_total0_); L.prof[32 + PH_TOTAL] += 1; }
#endif
}

// a wave that has found the dispenser empty: take open offers of the workgroup until no wave can draw a frame any more
LHIP_DEV void tail_help(const Tables& T, const PowBase& pb10, const Workspace& W, const StreamDesc* SD, int lane, QuantLds* Lall, int wv, int nwaves,
                        const QuantTabs& Q, TailShare& TS) {
    lane = lane_anew(lane);
    QuantLds& L = Lall[wv];
    const int C = T.channels_out, Cp = T.psy_channels;
    wg_add(&TS.drawing, -1, lane);
    long idle_rounds = 0;
    for (;;) {
        int took = 0;
        for (int w = 0; w < nwaves; w++) {
            if (w == wv) continue;
            TailOffer& o = TS.offer[w];
            if (wg_load(&o.state, lane) != 1) continue;
            if (wg_cas(&o.state, 1, 2, lane) != 1) continue;
            wg_acquire();
            took = 1;
            lane = lane_anew(lane);
            LHIP_TAIL_COUNT(claimed);
            const int fslot = uni(o.fslot), gr = uni(o.gr), mode_ext = uni(o.mode_ext), targ = uni(o.targ), gr0_bt = uni(o.gr0_bt);
            Seed seed; seed.start = uni(o.seed_start); seed.step = uni(o.seed_step);
            const int st = uni(W.fslot_stream[fslot]);
            const StreamDesc sd = SD[st];
            const int k = uni(fslot - sd.fslot0 - 1), fidx = uni(sd.out_slot0 + k), gslot = uni(sd.gslot0 + 1 + T.mode_gr * k + gr);
            const double ath_adjust = unid(W.ath_adjust[fslot]);
            q_ath_pseudo(T, pb10, ath_adjust, lane, L, Q);
            if (gr == 1) { LHIP_LANE_ONCE(i, 0, (SFBMAX) + 1) L.sf_gr0[1][i] = Lall[w].sf_gr0[1][i]; }
            wave_sync();
            const UnitOut u = q_unit(T, pb10, W, C, Cp, fidx, gslot, gr, 1, mode_ext, ath_adjust, targ, seed, gr0_bt, lane, L, Q);
            if (gr == 0) { LHIP_LANE_ONCE(i, 0, (SFBMAX) + 1) Lall[w].sf_gr0[1][i] = L.sf_gr0[1][i]; }
            if (lane == 0) { o.bits = u.bits; o.active = u.active; o.next_start = u.next.start; o.next_step = u.next.step; o.block_type = u.block_type; }
            wg_store(&o.state, 3, lane);
        }
        if (!took) {
            if (wg_load(&TS.drawing, lane) <= 0) break;
            wg_idle();
            LHIP_SPIN_GUARD(idle_rounds);
        }
    }
}

}  // namespace lhip
