/*
 * isg_walk_plan.h -- host side of the walk engine (isg_walk.h): how a run of groups is cut into segments, super-blocks and
 * blocks, how wide every block's window is, and where its table and maps live.  Sizes only: where a window SITS (its first
 * column) is worked out on the device from the shapes (wk_centers).
 */
#ifndef ISG_WALK_PLAN_H
#define ISG_WALK_PLAN_H
#include <math.h>
#include <vector>
#include "isg_walk.h"

#define WK_LDS_BUDGET (144 * 1024) /* per workgroup: tables of a block, maps of a super-block, maps of a segment */

struct WkPlan {
	std::vector<WkBlock> blk;
	std::vector<WkSuper> sup;
	std::vector<WkSeg> seg;
	std::vector<int> hw;      /* per block: half-width of its entry window */
	size_t table_bytes = 0;   /* of the largest segment (reuse = true) or of all of them */
	size_t maps_elems = 0;    /* 16-bit deltas, likewise */
	size_t lds_table = 0, lds_block = 0, lds_compose = 0, lds_top = 0;
	int groups = 0, ngmax = 0;
	double rho_hi = 0, sigma = 0, kwin = 0;
	bool reuse = true;
};

/*
 * gam0[G + 1]: first gamma of every group.  rho_hi: upper estimate of the rejected attempts per gamma (room for the growth inside
 * a block), sigma: their spread per gamma, kwin: half-width of an entry window in sigma sqrt(gammas since the segment's entry).
 * seg_groups: groups per segment at most (more segments: narrower windows, less table to build, more launches).
 * reuse: segments are processed one after the other and may share the table / map buffers.
 * false: the run does not fit the engine's limits (the caller uses its sequential path).
 */
/* entry_slack (interval mode, segments after the first): a later walk over the same tables may enter a segment this far from where the
 * walk that built them did */
static bool wk_plan_build(WkPlan &P, const int *gam0, int G, double rho_hi, double sigma, double kwin, int seg_groups, bool reuse, int mode, int entry_slack = 0)
{
	P = WkPlan();
	P.groups = G;
	P.rho_hi = rho_hi;
	P.sigma = sigma;
	P.kwin = kwin;
	P.reuse = reuse;
	if (G < 1) return false;
	for (int g = 0; g < G; g++) {
		const int n = gam0[g + 1] - gam0[g];
		if (n < 1 || n > WK_NGMAX) return false;
		if (n > P.ngmax) P.ngmax = n;
	}
	if (seg_groups < 64) seg_groups = 64;
	size_t toff_all = 0, foff_all = 0;
	int g = 0;
	while (g < G) {
		if ((int)P.seg.size() >= WK_MAXSEG) return false;
		int sg = (G - g < seg_groups) ? G - g : seg_groups;
		if (G - g > seg_groups) { /* the rest in equal parts (windows grow with the square root of a segment's length) */
			const int parts = (G - g + seg_groups - 1) / seg_groups;
			sg = (((G - g + parts - 1) / parts) + 63) & ~63;
			if (sg > seg_groups) sg = seg_groups;
		}
		/* this segment with blocks of bg groups; shrink bg until a block's table fits the LDS, the segment until its maps do */
		for (;;) {
			const int g1 = g + sg;
			const long gbase = gam0[g];
			/* widest window of the segment decides bg */
			const double nall = (double)(gam0[g1] - gbase);
			const int hmax = (int)ceil(kwin * sigma * sqrt(nall)) + 8 + ((g > 0) ? entry_slack : 0);
			int bg = WK_BG;
			for (;;) {
				const double nin = (double)P.ngmax * bg;
				const int grow = (int)ceil(rho_hi * nin + kwin * sigma * sqrt(nin)) + 16;
				const int W = ((2 * hmax + 1 + grow) + 63) & ~63;
				if ((size_t)bg * W <= WK_LDS_BUDGET || bg <= 8) break;
				bg /= 2;
			}
			std::vector<WkBlock> blk;
			std::vector<int> hw;
			size_t toff = reuse ? 0 : toff_all, foff = reuse ? 0 : foff_all;
			bool ok = true;
			size_t lds_block = 0, lds_table = 0;
			for (int b0 = g; b0 < g1; b0 += bg) {
				WkBlock B;
				B.g0 = b0;
				B.ng = (g1 - b0 < bg) ? g1 - b0 : bg;
				const double nbefore = (double)(gam0[b0] - gbase), nin = (double)(gam0[b0 + B.ng] - gam0[b0]);
				const int h = (b0 == g) ? ((g > 0) ? entry_slack : 0) : (int)ceil(kwin * sigma * sqrt(nbefore)) + 8 + ((g > 0) ? entry_slack : 0);
				const int grow = (int)ceil(rho_hi * nin + kwin * sigma * sqrt(nin)) + 16;
				B.wlo = 0;
				B.ein = 2 * h + 1;
				B.W = ((B.ein + grow) + 63) & ~63;
				B.toff = toff;
				B.foff = (unsigned)foff;
				toff += (size_t)B.ng * B.W;
				foff += (size_t)WK_EIN8(B.ein);
				if ((size_t)B.ng * B.W > WK_LDS_BUDGET) ok = false;
				if ((size_t)B.ng * B.W > lds_block) lds_block = (size_t)B.ng * B.W;
				const size_t lt = wk_table_lds_bytes(B.W, P.ngmax, mode);
				if (lt > lds_table) lds_table = lt;
				if (lt > WK_LDS_BUDGET) ok = false;
				blk.push_back(B);
				hw.push_back(h);
			}
			/* super-blocks and their maps */
			std::vector<WkSuper> sup;
			size_t lds_compose = 0, lds_top = 0;
			for (size_t k = 0; k < blk.size(); k += WK_FAN) {
				WkSuper S;
				S.b0 = (int)(P.blk.size() + k);
				S.nb = (int)((blk.size() - k < WK_FAN) ? blk.size() - k : WK_FAN);
				S.foff = (unsigned)foff;
				S.pad = 0;
				foff += (size_t)WK_EIN8(blk[k].ein);
				size_t lc = 0;
				for (int q = 0; q < S.nb; q++) lc += 2 * (size_t)WK_EIN8(blk[k + q].ein);
				if (lc > lds_compose) lds_compose = lc;
				lds_top += 2 * (size_t)WK_EIN8(blk[k].ein);
				sup.push_back(S);
			}
			if (lds_compose > WK_LDS_BUDGET) ok = false;
			if (lds_top > WK_LDS_BUDGET || !ok) {
				if (sg <= 64) return false;
				sg = (sg / 2 + 63) & ~63;
				if (sg < 64) sg = 64;
				continue;
			}
			if (foff >= 0xffffffffull) return false;
			WkSeg S;
			S.g0 = g;
			S.g1 = g1;
			S.b0 = (int)P.blk.size();
			S.nb = (int)blk.size();
			S.s0 = (int)P.sup.size();
			S.ns = (int)sup.size();
			S.bg = bg;
			S.pad = 0;
			P.seg.push_back(S);
			P.blk.insert(P.blk.end(), blk.begin(), blk.end());
			P.hw.insert(P.hw.end(), hw.begin(), hw.end());
			P.sup.insert(P.sup.end(), sup.begin(), sup.end());
			if (reuse) {
				if (toff > P.table_bytes) P.table_bytes = toff;
				if (foff > P.maps_elems) P.maps_elems = foff;
			} else {
				toff_all = toff;
				foff_all = foff;
				P.table_bytes = toff;
				P.maps_elems = foff;
			}
			if (lds_block > P.lds_block) P.lds_block = lds_block;
			if (lds_table > P.lds_table) P.lds_table = lds_table;
			if (lds_compose > P.lds_compose) P.lds_compose = lds_compose;
			if (lds_top > P.lds_top) P.lds_top = lds_top;
			g = g1;
			break;
		}
	}
	return true;
}

#endif
