// fa_plan64.hpp -- the compile-time filler plans of the persistent, hand-placed kernel (fa_fwd_kernel64.hpp;
// DESIGN.md 3.5): what rides in the gap behind every MFMA of a visit, and the checks that a plan keeps the wait-state
// distances the inline-asm MFMAs need (hipcc neither schedules nor pads them).  Host-evaluable constexpr, no device code:
// a placement is changed by changing one function here, and `static_assert(plan64_ok(...))` in the kernel refuses a plan
// that would read an accumulator too early.  What was tried and measured is in HISTORY.md 0 ("placement" rows): the
// barrier at the visit's top, DMA pieces late in phase 2, units spread evenly over both phases, 20 / 23 units in phase 1
// and the rotated plan's other depths were built as variants of these functions through round 5 and are gone from the tree.
#pragma once

namespace fa {

// What rides in the gap after MFMA g (g = 0..63 of a visit of the 64-rows-per-wave form: 0..31 = QK^T of tile it+1,
// 32..63 = P.V of tile it; the one-Q-tile-per-wave form uses gaps 0..31: 0..15 / 16..31).
struct Plan64 {
    signed char exp_first[64], exp_n[64];  // softmax units (2 elements each), 32 per visit, in P.V order
    signed char max_first[64], max_n[64];  // row-max units over S(it+1), 32 per visit
    signed char dma[64];                   // DMA piece 0..7 (0..3: the wave's four pieces of the K tile, 4..7: of the V tile) or -1
    signed char tail[64];                  // end-of-visit chain step 1.. or 0
    signed char barrier[64];               // 1: the visit's counted DMA wait + workgroup barrier
    signed char early_first[64], early_n[64];  // rotated plan: softmax units of the NEXT tile (S(it+1)), first rot_k of its 32
};

// The 64-rows-per-wave plan.  masked: gaps 32..35 of a diagonal visit rewrite S(it+1) (causal mask) before its row max is
// taken, so the 32 row-max units start 4 gaps later and the end-of-visit chain runs in 5 merged steps.  nomax (the
// speculative schedule): no row-max units, no end-of-visit chain -- only its last step, the next request pointers.
// n_phase1: softmax units of tile `it` that ride in phase 1 (the QK^T MFMAs), the rest follow in the odd gaps of phase 2.
// rot_k (speculative schedule only, DESIGN.md 3.5 "rotated units"): the last gaps of a visit run at the matrix pipe's own
// rate with issue slots to spare (the P.V MFMAs of the last 16-key slice; every unit of tile `it` has to be done two gaps
// before its slice is consumed, i.e. by gap 54), while phase 1 is issue bound.  So the first rot_k units of the NEXT tile
// -- S(it+1) is complete from gap 32 on, and P's slice 0 registers are free once gap 40 has issued -- ride in those last
// gaps, and a visit carries units rot_k .. 31 of its own tile + units 0 .. rot_k-1 of the next.
constexpr Plan64 make_plan64(bool masked, bool nomax, int n_phase1, int rot_k = 0) {
    Plan64 p{};
    const int m0 = masked ? 4 : 0, odd0 = masked ? 11 : 9, mend = masked ? 27 : 24;
    int e = rot_k, m = 0, d = 0, ee = 0;
    const int e1_end = rot_k + n_phase1;   // phase 1 carries units rot_k .. e1_end - 1
    // early units, one per gap: the odd gaps 55 .. 63 first (no operand reads there), then the even ones
    constexpr int early_order[10] = {55, 57, 59, 61, 63, 54, 56, 58, 60, 62};
    for (int g = 0; g < 64; ++g) {
        const int h = g - 32;
        int ne = 0, nm = 0, dm = -1, tl = 0;
        if (g < 32) {
            // gaps g % 4 == 0 carry the operand wait + two K reads; gap 2 the barrier; the DMA
            // pieces follow it, one per four gaps
            if ((g & 3) == 0) {
                if (g >= 4) dm = d++;
            } else if (e < e1_end && g != 2) {
                const int slot = (g >> 2) * 3 + (g & 3) - 1;          // 0..23 over the gaps with g % 4 != 0
                if ((slot + 1) * n_phase1 / 24 > slot * n_phase1 / 24) ne = 1;
            }
        } else {
            if (d < 8 && h == 0) dm = d++;                            // the eighth piece
            if ((h & 1) && h <= 21 && e < 32) {                       // odd gaps up to 53: the rest of the units
                const int gaps_left = (21 - h) / 2 + 1;
                ne = (32 - e + gaps_left - 1) / gaps_left;            // 1, or 2 while behind
            }
            if (h >= m0 && h < mend) {
                if (!(h & 1)) nm = 2;                                 // even gaps (with the V reads): 2
                else if (h >= odd0) nm = 1;                           // late odd gaps: 1  -> 24 + 8 = 32
            }
            if (masked) { if (h >= 27) tl = 10 + (h - 27); }          // merged chain steps 10..14
            else if (h >= 24) tl = h - 23;                            // chain steps 1..8
            if (nomax) { nm = 0; tl = (g == 63) ? 8 : 0; }
        }
        int nearly = 0;
        for (int i = 0; i < rot_k && i < 10; ++i) nearly += early_order[i] == g ? 1 : 0;
        p.early_first[g] = (signed char)ee; p.early_n[g] = (signed char)nearly; ee += nearly;
        p.exp_first[g] = (signed char)e; p.exp_n[g] = (signed char)ne; e += ne;
        p.max_first[g] = (signed char)m; p.max_n[g] = (signed char)nm; m += nm;
        p.dma[g] = (signed char)dm; p.tail[g] = (signed char)tl;
        p.barrier[g] = g == 2 ? 1 : 0;
    }
    return p;
}
constexpr bool plan64_ok(const Plan64 &p, int rot_k = 0) {
    int e = rot_k, m = 0, d = 0, bar = -1, ee = 0;
    bool chain = false;  // the plan carries the row max + the end-of-visit chain (not the speculative schedule)
    for (int g = 0; g < 64; ++g) {
        // P of 16-key slice s16 is consumed from gap 32 + 8 s16 on: its units must be >= 2 gaps older
        for (int u = p.exp_first[g]; u < p.exp_first[g] + p.exp_n[g]; ++u)
            if (g + 2 > 32 + 8 * (u >> 3)) return false;
        // an early unit (of the next tile) writes P's slice u >> 3, last read by the MFMA of gap 32 + 8 (u >> 3) + 7 (whose
        // operands stay allocated until the next MFMA has issued), and reads S(it+1), complete two MFMAs behind gap 31
        for (int u = p.early_first[g]; u < p.early_first[g] + p.early_n[g]; ++u)
            if (g < 32 + 8 * (u >> 3) + 9 || g < 34 || u >= 16) return false;
        ee += p.early_n[g];
        // S(it+1) tiles: nt = 0 last written at gap 29, nt = 1 at gap 31; read >= 2 MFMAs later
        for (int u = p.max_first[g]; u < p.max_first[g] + p.max_n[g]; ++u)
            if (g < ((u >> 4) ? 34 : 32)) return false;
        if ((p.tail[g] == 1 || p.tail[g] == 10) && m < 32) return false;
        if (p.tail[g] == 1 || p.tail[g] == 10) chain = true;
        if (p.barrier[g]) bar = g;
        if (p.dma[g] >= 0 && (bar < 0 || g <= bar)) return false;     // DMA overwrites what the barrier frees
        if (p.dma[g] >= 0 && g > 0 && p.dma[g - 1] >= 0) return false; // a DMA piece needs the gap before it for its M0
        if (p.barrier[g] && g >= 28) return false;                    // V(it+1) is first read at gap 30
        e += p.exp_n[g]; m += p.max_n[g]; d += p.dma[g] >= 0;
    }
    return e == 32 && ee == rot_k && m == (chain ? 32 : 0) && d == 8 && bar >= 0;
}

// One Q tile per wave (QTP = 1): a visit is 32 MFMAs, one gap per operand step -- gaps 0..15 S(it+1) = K(it+1) Q^T
// (step = 2 ks + nt), gaps 16..31 O += V(it) P(it) (step 16 + 4 s16 + t).  Per visit: 16 softmax units (u = 4 s16 + j),
// 16 row-max units over S(it+1) (complete behind gap 15), the merged end-of-visit chain (steps 10..14), the 8 DMA
// pieces at the even gaps 2..16 (each needs the gap before it for its M0), the barrier in gap 1; the operand wait and
// the two reads of the next pair sit in the even gaps.  Unit u's P slice s16 = u / 4 is consumed from gap 16 + 4 s16.
constexpr Plan64 make_plan32(bool nomax = false, int n_pieces = 8) {   // n_pieces: 8 (four waves), 4 (eight waves share a tile's 16 pieces)
    Plan64 p{};
    int e = 0, m = 0, d = 0;
    for (int g = 0; g < 32; ++g) {
        const int h = g - 16;
        int ne = 0, nm = 0, dm = -1, tl = 0;
        if ((g & 1) == 0 && g >= 2 && g <= 16 && d < n_pieces) dm = d++;
        if (g < 16) {
            // 11 units: the odd gaps 3..15, and the even gaps 4, 8, 12, 14 (the lighter ones: no DMA issue cost twice)
            if (g >= 3 && ((g & 1) || g == 4 || g == 8 || g == 12 || g == 14)) ne = 1;
        } else {
            if ((h & 1) && h <= 9) ne = 1;                       // units 11..15 at gaps 17, 19, 21, 23, 25
            if (h <= 10) nm = (!(h & 1) && h < 10) ? 2 : 1;      // 2 1 2 1 2 1 2 1 2 1 1 = 16
            if (h >= 11) tl = 10 + (h - 11);                     // merged chain steps 10..14
            if (nomax) { nm = 0; tl = (g == 31) ? 8 : 0; }       // speculative schedule: only the next request pointers
        }
        p.exp_first[g] = (signed char)e; p.exp_n[g] = (signed char)ne; e += ne;
        p.max_first[g] = (signed char)m; p.max_n[g] = (signed char)nm; m += nm;
        p.dma[g] = (signed char)dm; p.tail[g] = (signed char)tl;
        p.barrier[g] = g == 1 ? 1 : 0;
        p.early_first[g] = 0; p.early_n[g] = 0;
    }
    for (int g = 32; g < 64; ++g) { p.dma[g] = -1; }
    return p;
}
constexpr int plan_barrier_gap(const Plan64 &p, int n_gaps) {
    for (int g = 0; g < n_gaps; ++g)
        if (p.barrier[g]) return g;
    return -1;
}
constexpr bool plan32_ok(const Plan64 &p, bool nomax = false, int n_pieces = 8) {
    int e = 0, m = 0, d = 0, bar = -1;
    for (int g = 0; g < 32; ++g) {
        for (int u = p.exp_first[g]; u < p.exp_first[g] + p.exp_n[g]; ++u)
            if (g + 2 > 16 + 4 * (u >> 2)) return false;             // packed >= 2 gaps before its slice is consumed
        // S(it+1): nt = 0 last written at gap 14, nt = 1 at gap 15; read >= 2 MFMAs later
        for (int u = p.max_first[g]; u < p.max_first[g] + p.max_n[g]; ++u)
            if (g < ((u >> 3) ? 17 : 16)) return false;
        if (p.tail[g] == 10 && m < 16) return false;
        if (p.barrier[g]) bar = g;
        if (p.dma[g] >= 0 && (bar < 0 || g <= bar)) return false;    // DMA overwrites what the barrier frees
        if (p.dma[g] >= 0 && g > 0 && p.dma[g - 1] >= 0) return false;
        if (p.barrier[g] && g >= 28) return false;                   // K(it+2) is first read at gap 30
        e += p.exp_n[g]; m += p.max_n[g]; d += p.dma[g] >= 0;
    }
    return e == 16 && m == (nomax ? 0 : 16) && d == n_pieces && bar >= 0;
}

}  // namespace fa
