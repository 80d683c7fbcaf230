// nd_build.h -- host-side symbolic phase of the nested-dissection pressure solve (grid only: shared by every member and
// every time step).  Plain C++; used by press_nd.hip and, through hm_debug_nd_tables, by the CPU tests.
#pragma once
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "nd.h"

struct NdTablesHost {
    NdInfo info{};
    std::vector<int> fronts;   // n_fronts * ND_FRONT_INTS
    std::vector<int> cells;    // position -> cell (-1 padding, -2 right-hand side)
    std::vector<short> rec;    // assembly recipes (nd.h), n_rec_blocks * 256
    std::vector<short> cpos;   // per front: [child 0: 16 T entries][child 1: 16 T entries] at 2 * cells_off:
                               // position of the front position's cell in the child's boundary list (b_child = its
                               // right-hand-side row), -1 if the child's update has no such row
    std::string error;
};

namespace nd_detail {

struct Region {
    int x0, x1, y0, y1;
    bool side[4];  // W, E, S, N border an ancestor's separator
};

inline void region_cells(std::vector<int>& out, int Ny, int x0, int x1, int y0, int y1) {
    for (int x = x0; x < x1; ++x)
        for (int y = y0; y < y1; ++y) out.push_back(x * Ny + y);
}

struct Builder {
    int Nx, Ny, leaf, levels;
    std::vector<std::vector<int>> piv, bnd;  // per front id
    std::vector<int> level, c0, c1, rbox, rboy;
    bool complete = true;

    void dissect(const Region& r, int lv, int idx) {
        const int id = (1 << lv) - 1 + idx;
        if ((int)piv.size() <= id) {
            piv.resize(id + 1); bnd.resize(id + 1); level.resize(id + 1, -1); c0.resize(id + 1, -1); c1.resize(id + 1, -1); rbox.resize(id + 1, 0); rboy.resize(id + 1, 0);
        }
        level[id] = lv;
        rbox[id] = r.x0 | (r.x1 << 16);
        rboy[id] = r.y0 | (r.y1 << 16);
        std::vector<int>& b = bnd[id];
        if (r.side[0]) region_cells(b, Ny, r.x0 - 1, r.x0, r.y0, r.y1);
        if (r.side[1]) region_cells(b, Ny, r.x1, r.x1 + 1, r.y0, r.y1);
        if (r.side[2]) region_cells(b, Ny, r.x0, r.x1, r.y0 - 1, r.y0);
        if (r.side[3]) region_cells(b, Ny, r.x0, r.x1, r.y1, r.y1 + 1);
        const int w = r.x1 - r.x0, h = r.y1 - r.y0;
        if (w <= leaf && h <= leaf) {
            if (lv != levels - 1) complete = false;
            region_cells(piv[id], Ny, r.x0, r.x1, r.y0, r.y1);
            return;
        }
        if (lv >= levels - 1) { complete = false; return; }
        Region a = r, c = r;
        if (w >= h) {
            const int xs = r.x0 + (w - 1) / 2;
            region_cells(piv[id], Ny, xs, xs + 1, r.y0, r.y1);
            a.x1 = xs; a.side[1] = true;
            c.x0 = xs + 1; c.side[0] = true;
            if (xs <= r.x0 || r.x1 <= xs + 1) { complete = false; return; }
        } else {
            const int ys = r.y0 + (h - 1) / 2;
            region_cells(piv[id], Ny, r.x0, r.x1, ys, ys + 1);
            a.y1 = ys; a.side[3] = true;
            c.y0 = ys + 1; c.side[2] = true;
            if (ys <= r.y0 || r.y1 <= ys + 1) { complete = false; return; }
        }
        c0[id] = (1 << (lv + 1)) - 1 + 2 * idx;
        c1[id] = c0[id] + 1;
        dissect(a, lv + 1, 2 * idx);
        dissect(c, lv + 1, 2 * idx + 1);
    }
};

}  // namespace nd_detail

// Returns false (with t.error set) when the grid does not give the complete tree the kernels are written for (11 levels at 128 x 128,
// 13 at 256 x 256, 15 at 512 x 512).
inline bool nd_build_tables(int Nx, int Ny, NdTablesHost& t) {
    using namespace nd_detail;
    Builder B;
    B.Nx = Nx; B.Ny = Ny; B.leaf = 4;
    int lo = -1;
    if (Nx == Ny && (Nx == 128 || Nx == 256 || Nx == 512)) lo = Nx == 128 ? 0 : Nx == 256 ? 2 : 4;
    if (lo < 0) { t.error = "grid does not dissect into the complete tree the kernels are written for (128, 256 or 512 cells a side)"; return false; }
    const int levels = 11 + lo;
    B.levels = levels;
    Region root{0, Nx, 0, Ny, {false, false, false, false}};
    B.dissect(root, 0, 0);
    const int nF = (1 << levels) - 1;
    if (!B.complete || (int)B.piv.size() != nF) { t.error = "grid does not dissect into the complete tree"; return false; }
    {   // boundary cells of a front with children, ordered by child (nd.h): [child 0's | both (none on this grid) | neither's | child 1's]
        std::vector<unsigned char> mark((size_t)Nx * Ny, 0);
        for (int f = 0; f < nF; ++f) {
            if (B.c0[f] < 0) continue;
            for (int c : B.bnd[B.c0[f]]) mark[c] |= 1;
            for (int c : B.bnd[B.c1[f]]) mark[c] |= 2;
            static const int rank[4] = {2, 0, 3, 1};  // mark -> place
            std::stable_sort(B.bnd[f].begin(), B.bnd[f].end(), [&](int a, int c) { return rank[mark[a]] < rank[mark[c]]; });
            for (int c : B.bnd[B.c0[f]]) mark[c] = 0;
            for (int c : B.bnd[B.c1[f]]) mark[c] = 0;
        }
    }
    for (int f = 0; f < nF; ++f)
        if (B.level[f] < 0) { t.error = "missing front"; return false; }
    {  // every cell is a pivot exactly once
        std::vector<int> seen((size_t)Nx * Ny, 0);
        for (int f = 0; f < nF; ++f)
            for (int c : B.piv[f]) seen[c]++;
        for (int v : seen)
            if (v != 1) { t.error = "a cell is not a pivot exactly once"; return false; }
    }
    NdInfo& I = t.info;
    I = NdInfo{};
    I.n_fronts = nF;
    I.levels = levels;
    I.lo = lo;
    const int big_top = lo > 0 ? lo + 2 : -1;  // levels 0 .. big_top: big fronts (tile-format updates); levels lo + 3, lo + 4: k_nd_top's, packed
    t.fronts.assign((size_t)nF * ND_FRONT_INTS, 0);
    t.cells.clear();
    long long fact = 0, arena = 0, pimg = 0;
    std::vector<int> where((size_t)Nx * Ny, -1);
    for (int f = 0; f < nF; ++f) {
        int* F = &t.fronts[(size_t)f * ND_FRONT_INTS];
        const int lv = B.level[f], s = (int)B.piv[f].size(), b = (int)B.bnd[f].size();
        const int st = (s + 15) / 16, bt = (b + 1 + 15) / 16, T = st + bt;
        F[NDF_LEVEL] = lv; F[NDF_S] = s; F[NDF_B] = b; F[NDF_ST] = st; F[NDF_BT] = bt;
        F[NDF_C0] = B.c0[f]; F[NDF_C1] = B.c1[f];
        F[NDF_CELLS] = (int)t.cells.size();
        const int last = s - 16 * (st - 1);
        F[NDF_KREG] = (last + 3) / 4;
        F[NDF_FACT] = (int)fact;
        {
            int x0 = Nx, y0 = Ny, x1 = 0, y1 = 0;
            for (int c : B.piv[f]) { x0 = std::min(x0, c / Ny); x1 = std::max(x1, c / Ny + 1); y0 = std::min(y0, c % Ny); y1 = std::max(y1, c % Ny + 1); }
            F[NDF_PBOX] = x0 | (x1 << 16);
            F[NDF_PBOY] = y0 | (y1 << 16);
            F[NDF_RBOX] = B.rbox[f];
            F[NDF_RBOY] = B.rboy[f];
        }
        long long tiles_regs = 0;  // 64-double register rows
        for (int p = 0; p < st; ++p) tiles_regs += (long long)(T - p - 1) * (p == st - 1 ? F[NDF_KREG] : 4);
        if (lv < levels - 1) fact += tiles_regs * 64;  // (the leaves keep no factor: k_nd_leaf_solve eliminates them again)
        if (fact > 0x7fffffffLL) { t.error = "factor offset exceeds int32"; return false; }
        F[NDF_PIMG] = -1;
        if (lv <= big_top) { F[NDF_PIMG] = (int)pimg; pimg += (long long)st * 256; }
        if (lv == big_top) I.big_fact_doubles = fact;  // (fronts are numbered level by level: the last front of level big_top sets it last)
        long long upd_n = ((long long)(b + 1) * (b + 2) / 2 + 1) & ~1LL;
        if (lv <= big_top) upd_n = (long long)bt * (bt + 1) / 2 * 256;  // whole tiles (nd.h)
        if ((lv - lo <= ND_ARENA_MAX_LEVEL && lv > 0) || lv == levels - 1) {  // leaves: their updates go from k_nd_leaf to k_nd_sub through the arena
            F[NDF_UPD] = (int)arena;
            arena += upd_n;
            if (arena > 0x7fffffffLL) { t.error = "arena offset exceeds int32"; return false; }
        } else F[NDF_UPD] = -1;
        I.upd_doubles[lv] = std::max(I.upd_doubles[lv], (int)upd_n);
        I.max_bt[lv] = std::max(I.max_bt[lv], bt);
        I.max_st[lv] = std::max(I.max_st[lv], st);
        t.cells.resize(t.cells.size() + (size_t)16 * T, -1);
        int* C = &t.cells[F[NDF_CELLS]];
        for (int i = 0; i < s; ++i) C[i] = B.piv[f][i];
        for (int i = 0; i < b; ++i) C[16 * st + i] = B.bnd[f][i];
        C[16 * st + b] = -2;
    }
    I.n_cells = (int)t.cells.size();
    I.fact_doubles = fact;
    I.arena_doubles = arena;
    I.pimg_doubles = pimg;
    t.cpos.assign((size_t)2 * t.cells.size(), (short)-1);
    for (int f = 0; f < nF; ++f) {
        int* F = &t.fronts[(size_t)f * ND_FRONT_INTS];
        const int T = F[NDF_ST] + F[NDF_BT];
        const int* C = &t.cells[F[NDF_CELLS]];
        F[NDF_BC0] = F[NDF_BC1] = 0;
        for (int c = 0; c < 2; ++c) {
            const int ch = c == 0 ? F[NDF_C0] : F[NDF_C1];
            if (ch < 0) continue;
            const std::vector<int>& cb = B.bnd[ch];
            F[c == 0 ? NDF_BC0 : NDF_BC1] = (int)cb.size();
            F[c == 0 ? NDF_UC0 : NDF_UC1] = t.fronts[(size_t)ch * ND_FRONT_INTS + NDF_UPD];
            for (size_t i = 0; i < cb.size(); ++i) where[cb[i]] = (int)i;
            short* P = &t.cpos[(size_t)2 * F[NDF_CELLS] + (size_t)c * 16 * T];
            size_t found = 0;
            for (int p = 0; p < 16 * T; ++p) {
                if (C[p] >= 0 && where[C[p]] >= 0) { P[p] = (short)where[C[p]]; ++found; }
                else if (C[p] == -2) P[p] = (short)cb.size();
            }
            for (size_t i = 0; i < cb.size(); ++i) where[cb[i]] = -1;
            if (found != cb.size()) { t.error = "a child's boundary does not lie inside its parent's front"; return false; }
        }
        // which tile rows hold anything of child 0 / child 1 / a matrix coefficient against a pivot (nd.h: NDF_KIDM, NDF_COFM)
        F[NDF_KIDM] = F[NDF_COFM] = -1;
        if (T <= 16 && F[NDF_C0] >= 0) {
            const int st = F[NDF_ST], s = F[NDF_S];
            const short* P0 = &t.cpos[(size_t)2 * F[NDF_CELLS]];
            const short* P1 = P0 + 16 * T;
            int km = 0, cm = 0;
            for (int R = 0; R < T; ++R)
                for (int i = 0; i < 16; ++i) {
                    const int p = 16 * R + i;
                    if (P0[p] >= 0) km |= 1 << R;
                    if (P1[p] >= 0) km |= 0x10000 << R;
                    bool co = R < st || C[p] == -2;
                    if (C[p] >= 0)
                        for (int k = 0; k < s && !co; ++k) {
                            const int d = C[p] - C[k];
                            co = d == 0 || d == Ny || d == -Ny || (d == 1 && C[k] % Ny != Ny - 1) || (d == -1 && C[k] % Ny != 0);
                        }
                    if (co) cm |= 1 << R;
                }
            F[NDF_KIDM] = km;
            F[NDF_COFM] = cm;
        }
    }
    // ---- assembly recipes (nd.h)
    t.rec.clear();
    auto tri = [](int a, int c) { const int hi = a > c ? a : c, lo = a > c ? c : a; return hi * (hi + 1) / 2 + lo; };
    auto coef_local = [&](int cm, int ck, bool same_pos, int box, int boy, int plane) -> int {  // offset into the staged LDS planes
        if (ck < 0) return same_pos ? -2 : -1;
        const int x0 = box & 0xffff, y0 = boy & 0xffff, y1 = boy >> 16, ld = y1 - y0 + 2;
        const int li = (ck / Ny - x0 + 1) * ld + (ck % Ny - y0 + 1);
        if (cm == -2) return 3 * plane + li;
        if (cm < 0) return -1;
        const int d = cm - ck;
        if (d == 0) return li;
        if (d == Ny) return plane + li + ld;
        if (d == -Ny) return plane + li;
        if (d == 1 && ck % Ny != Ny - 1) return 2 * plane + li + 1;
        if (d == -1 && ck % Ny != 0) return 2 * plane + li;
        return -1;
    };
    for (int f = 0; f < nF; ++f) {
        int* F = &t.fronts[(size_t)f * ND_FRONT_INTS];
        const int lv = F[NDF_LEVEL] - lo, b = F[NDF_B], st = F[NDF_ST], bt = F[NDF_BT], T = st + bt;  // lv: the level in the 128 x 128 tree's numbering
        const int* C = &t.cells[F[NDF_CELLS]];
        const short* P0 = &t.cpos[(size_t)2 * F[NDF_CELLS]];
        const short* P1 = P0 + 16 * T;
        const bool kids = F[NDF_C0] >= 0;
        F[NDF_REC] = lv >= 5 ? (int)(t.rec.size() / 256) : -1;
        auto block = [&]() -> size_t { t.rec.resize(t.rec.size() + 256, (short)-1); return t.rec.size() - 256; };
        // (levels >= 5: `base` >= 0 makes the entry a BYTE offset into the wave's LDS block, nd.h, with "none" -> the zero cell)
        auto gather_block = [&](const short* P, int rowpos0, int colpos0, int base = -1) {
            // entry (lane, r): row position rowpos0 + 4 r + lq, column position colpos0 + lc  (positions in the front)
            short* B = &t.rec[block()];
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {
                    const int a = P[rowpos0 + 4 * r + (lane >> 4)], c = P[colpos0 + (lane & 15)];
                    int v = (a >= 0 && c >= 0) ? tri(a, c) : -1;
                    if (base >= 0) v = 8 * (v >= 0 ? base + v : ND_LDS_ZERO);
                    if (v > 32767) { t.error = "recipe offset exceeds int16"; v = -1; }
                    B[lane * 4 + r] = (short)v;
                }
        };
        auto out_block = [&](int R, int Cc) {  // R, Cc: boundary tile rows counted from 0
            short* B = &t.rec[block()];
            for (int lane = 0; lane < 64; ++lane)
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * R + 4 * r + (lane >> 4), j = 16 * Cc + (lane & 15);
                    const int v = (j <= i && i <= b) ? i * (i + 1) / 2 + j : -1;
                    B[lane * 4 + r] = (short)v;
                }
        };
        if (lv >= 5) {
            int box = F[NDF_PBOX], boy = F[NDF_PBOY], plane = ND_CF_PLANE_WAVE;
            // the wave's LDS block (nd.h): where the children's updates and the coefficient planes lie
            int c0base, c1base, cfbase;
            if (lv >= 8) {
                const int idx = f - ((1 << (lv + lo)) - 1), f8 = (1 << (8 + lo)) - 1 + (idx >> (lv - 8));
                box = t.fronts[(size_t)f8 * ND_FRONT_INTS + NDF_RBOX];
                boy = t.fronts[(size_t)f8 * ND_FRONT_INTS + NDF_RBOY];
                plane = ND_CF_PLANE_SUB;
                const int s9 = I.upd_doubles[9 + lo], s10 = I.upd_doubles[10 + lo];
                c0base = lv == 8 ? ND_LDS_DATA : ND_LDS_DATA + 2 * s9;            // level 8 reads the level-9 slots, level 9 the level-10 slots
                c1base = c0base + (lv == 8 ? s9 : s10);
                cfbase = ND_LDS_DATA + 2 * (s9 + s10);
            } else {
                const int chd = I.upd_doubles[lv + lo + 1];
                c0base = ND_LDS_DATA;
                c1base = ND_LDS_DATA + chd;
                cfbase = ND_LDS_DATA + 2 * chd;
            }
            {   // the staged planes must hold the box plus its ring
                const int x0 = box & 0xffff, y0 = boy & 0xffff, x1 = box >> 16, y1 = boy >> 16;
                if ((x1 - x0 + 2) * (y1 - y0 + 2) > plane) { t.error = "coefficient box larger than its LDS plane"; return false; }
            }
            for (int R = 0; R <= bt; ++R) {
                short* B = &t.rec[block()];
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        const int k = 4 * r + (lane >> 4), m = 16 * R + (lane & 15);
                        const int v = coef_local(C[m], C[k], k == m, box, boy, plane);
                        const int off = 8 * (v >= 0 ? cfbase + v : (v == -2 ? ND_LDS_ONE : ND_LDS_ZERO));
                        if (off > 32767) { t.error = "recipe offset exceeds int16"; return false; }
                        B[lane * 4 + r] = (short)off;
                    }
                if (kids) {
                    gather_block(P0, 0, 16 * R, c0base);
                    gather_block(P1, 0, 16 * R, c1base);
                }
            }
            for (int R = 1; R <= bt; ++R)
                for (int Cc = 1; Cc <= R; ++Cc) {
                    if (kids) {
                        gather_block(P0, 16 * R, 16 * Cc, c0base);
                        gather_block(P1, 16 * R, 16 * Cc, c1base);
                    }
                    out_block(R - 1, Cc - 1);
                }
        }
    }
    if (!t.error.empty()) return false;
    I.n_rec_blocks = (int)(t.rec.size() / 256);
    return true;
}
