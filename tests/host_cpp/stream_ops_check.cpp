// Host-side check of the tables the fused kernels assemble a view's record with (kernels.hpp: buildEmitTable,
// buildStreamOps). Compiled and run by tests/test_host_cpu.py with hipcc; only host functions run -- no GPU needed:
//  * every record entry the per-view kernels read -- the six view rows of J^T J and the view's six entries of J^T r --
//    is stored by exactly ONE (lane, op) of the stream form's direct emission, nothing else is stored, and every
//    offset is either inside the 768-byte record or the "no op" offset the buffer range check drops; the six slots of
//    V's upper triangle that carry g_v (gvSlot) take J^T r, every other slot the J^T J entry of its row and column;
//  * an op's source matches the table entry it serves: a sum of both row kinds (ops 0-5) exactly where the emit table
//    adds the u tile and the v tile at the same index, a single part (ops 6-10) where it picks one.
#include "../../camera-calibration_amd/csrc/kernels.hpp"
#include <cstdio>
#include <map>
#include <set>

using namespace calib;

static int check(int C) {
    const int L = C - 6;
    uint32_t tab[kEmitTabSize];
    buildEmitTable(C, tab);
    int32_t ops[64 * kStreamOps];
    if (!buildStreamOps(C, ops)) { std::printf("C=%d: buildStreamOps reports an inconsistency\n", C); return 1; }
    std::set<int> wanted;
    int bad = 0;
    for (int slot = 0; slot < kGStride; ++slot) {
        const int iu = (int)(tab[slot] & 0xffff), iv = (int)(tab[slot] >> 16);
        if (!(iu == kEmitZero && iv == kEmitZero)) wanted.insert(slot);
        // g_v[j] sits where gvSlot says, round trip through gvOfSlot, strictly above V's diagonal
        const int j = gvOfSlot(L, slot);
        if (j >= 0) {
            const int m = slot / 16, n = slot % 16 - L;
            if (gvSlot(L, j) != slot || n <= m || n > 5) { std::printf("C=%d: slot %d is not a free slot of V for g_v[%d]\n", C, slot, j); ++bad; }
            uint32_t ref[kEmitTabSize];
            buildEmitTable(C, ref);
            // J^T r entry L + j: the residual's row of the tiles (fisheye: row 15; radtan: column 4 of row L + j)
            const int want = C == 15 ? 15 * 16 + L + j : (L + j) * 16 + 4;
            if (iu != want || iv != want) { std::printf("C=%d: slot %d does not take (J^T r)[%d]\n", C, slot, L + j); ++bad; }
        }
    }
    for (int j = 0; j < 6; ++j)
        if (gvOfSlot(L, gvSlot(L, j)) != j) { std::printf("C=%d: gvSlot / gvOfSlot disagree for %d\n", C, j); ++bad; }
    std::map<int, int> writers;
    for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < kStreamOps; ++j) {
            const int off = ops[lane * kStreamOps + j];
            if (off == kStreamNoOp) continue;
            if (off < 0 || off >= kGStride * 8 || (off & 7)) { std::printf("C=%d lane %d op %d: offset %d\n", C, lane, j, off); ++bad; continue; }
            const int slot = off / 8;
            writers[slot] += 1;
            if (!wanted.count(slot)) { std::printf("C=%d lane %d op %d stores slot %d nobody reads\n", C, lane, j, slot); ++bad; }
            const int iu = (int)(tab[slot] & 0xffff), iv = (int)(tab[slot] >> 16);
            const bool both = iu != kEmitZero && iv != kEmitZero;
            if (both != (j < 6)) { std::printf("C=%d lane %d op %d (slot %d): source kind does not match the emit table\n", C, lane, j, slot); ++bad; }
            if (j == 11) { std::printf("C=%d lane %d uses the spare op 11\n", C, lane); ++bad; }
            if (C == 15 && j >= 6) { std::printf("C=15 lane %d uses single-part op %d (the kernel skips them for fisheye)\n", lane, j); ++bad; }
        }
    for (int slot : wanted)
        if (writers[slot] != 1) { std::printf("C=%d: slot %d has %d writers\n", C, slot, writers[slot]); ++bad; }
    std::printf("C=%d: %zu record entries, each stored by exactly one lane: %s\n", C, wanted.size(), bad ? "FAILED" : "ok");
    return bad;
}

// stream_extra_item (which overflow record, if any, holds the part of view v a second wave summed) against a direct
// simulation of the wave cuts: wave w starts at group w * share; a start strictly inside a view cuts that view.
static int checkCuts() {
    int bad = 0, cases = 0;
    for (int n4 = 16; n4 <= 128; n4 += (n4 < 32 ? 1 : 7))
        for (int nv = 1; nv <= 70; nv += (nv < 12 ? 1 : 9))
            for (int waves = 1; waves <= nv; ++waves) {
                const long long groups = (long long)nv * n4;
                const int share = (int)((groups + waves - 1) / waves);
                std::map<int, int> extra;
                for (long long w = 1; w * share < groups; ++w)
                    if ((w * share) % n4 != 0) extra[(int)(w * share / n4)] = nv + (int)w;
                StreamMap sm; sm.share = share; sm.n4 = n4; sm.nv = nv;
                for (int v = 0; v < nv; ++v) {
                    const int want = extra.count(v) ? extra[v] : -1, got = stream_extra_item(sm, v);
                    if (want != got) { if (bad < 10) std::printf("n4=%d nv=%d waves=%d share=%d view %d: %d, simulation %d\n", n4, nv, waves, share, v, got, want); ++bad; }
                }
                ++cases;
            }
    StreamMap off; off.share = 0; off.n4 = 50; off.nv = 10;
    if (stream_extra_item(off, 3) != -1) ++bad;
    std::printf("wave cuts: %d (views per group count, views, waves) cases: %s\n", cases, bad ? "FAILED" : "ok");
    return bad;
}

int main() { return (check(15) + check(16) + checkCuts()) ? 1 : 0; }
