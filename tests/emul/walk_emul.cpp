/*
 * walk_emul.cpp -- TEST INFRASTRUCTURE.  Runs the walk engine's kernel bodies (instruct_amd/csrc/isg_walk.h: the very source the
 * device compiles) on the host, workgroup by workgroup and phase by phase, next to the plain sequential sampler
 * (isg_sampler.h: rdirich as random.c:264-280 draws it).  No product path links this.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../instruct_amd/csrc/isg_walk_plan.h"

extern "C" unsigned wk_ref_consumed(unsigned long long pos, const double *shape, int n, long s1, long s2, long s3);
#include <stddef.h>
static void run_segment(const WkPlan &P, int s, std::vector<WkBlock> &blk, const int *gam0, const unsigned long long *gpos, const int *gcnt, const float *alo, const float *ahi,
			std::vector<unsigned char> &table, std::vector<unsigned short> &maps, std::vector<int> &ent_sup, std::vector<int> &ent_blk,
			unsigned long long *T, WkState *st, isg_wh base, const isg_wh_tables *tab, int mode, int strict, float scale, int nthreads, bool build_table, int keep_origin = 0)
{
	const WkSeg S = P.seg[s];
	std::vector<unsigned char> lds(160 * 1024);
	static std::vector<float> bpred;
	if (bpred.size() < P.blk.size()) bpred.resize(P.blk.size());
	if (build_table) {
		WkCenterArgs C;
		C.bpred = bpred.data(); C.st = st;
		C.blk = blk.data(); C.hw = P.hw.data(); C.gam0 = gam0; C.gcnt = gcnt; C.alo = alo; C.ahi = ahi; C.seg = S; C.mode = mode; C.scale = scale;
		C.segs = nullptr;
		for (int wg = 0; wg < S.nb; wg++) wk_bpred_body(C, S.b0 + wg, 64, lds.data());
		wk_centers_body(C, 0, nthreads, lds.data());
		WkTableArgs A;
		A.blk = blk.data(); A.gam0 = gam0; A.gpos = gpos; A.gcnt = gcnt; A.alo = alo; A.ahi = ahi; A.table = table.data(); A.st = st; A.base = base; A.tab = tab; A.tape = nullptr; A.tape_len = 0;
		A.seg_b0 = S.b0; A.seg_g0 = S.g0; A.bg = S.bg; A.seg = s; A.mode = mode;
		for (int wg = 0; wg < S.g1 - S.g0; wg++) wk_table_body(A, wg, nthreads, lds.data());
	}
	WkWalkArgs W;
	W.blk = blk.data(); W.sup = P.sup.data(); W.table = table.data(); W.maps = maps.data(); W.ent_sup = ent_sup.data(); W.ent_blk = ent_blk.data(); W.T = T; W.st = st;
	W.bpred = bpred.data(); W.gam0 = gam0; W.scale = scale;
	W.seg = S; W.segno = s; W.mode = mode; W.strict = strict; W.keep_origin = keep_origin; W.total_groups = P.groups;
	for (int wg = 0; wg < S.nb; wg++) wk_block_body(W, wg, nthreads, lds.data());
	for (int wg = 0; wg < S.ns; wg++) wk_compose_body(W, wg, nthreads, lds.data());
	wk_top_body(W, nthreads, lds.data());
	for (int wg = 0; wg < S.ns; wg++) wk_expand_body(W, wg, nthreads, lds.data());
	for (int wg = 0; wg < S.nb; wg++) wk_final_body(W, wg, nthreads, lds.data());
}

extern "C" int wk_emul_exact(const int *gam0, int G, const int *gcnt, long s1, long s2, long s3, double rho_hi, double sigma, double kwin, int seg_groups, float scale,
			     int nthreads, unsigned long long *T, unsigned long long *out /* [8]: total uniforms, fail, namb, blocks, segments, table bytes, sum_d, sum_d2 */)
{
	WkPlan P;
	if (!wk_plan_build(P, gam0, G, rho_hi, sigma, kwin, seg_groups, true, WK_MODE_EXACT)) return 1;
	const int NG = gam0[G];
	std::vector<unsigned long long> gpos(NG + 1, 0);
	for (int k = 0; k < NG; k++) gpos[k + 1] = gpos[k] + (gcnt[k] == 0 ? 1u : 2u);
	isg_wh_tables tab;
	isg_wh_tables_init(&tab);
	isg_wh base;
	base.s1 = (uint32_t)(s1 % ISG_M1); base.s2 = (uint32_t)(s2 % ISG_M2); base.s3 = (uint32_t)(s3 % ISG_M3);
	std::vector<WkBlock> blk = P.blk;
	std::vector<unsigned char> table(P.table_bytes + 64);
	std::vector<unsigned short> maps(P.maps_elems + 64);
	std::vector<int> ent_sup(P.sup.size()), ent_blk(P.blk.size());
	WkState st;
	memset(&st, 0, sizeof(st));
	for (size_t s = 0; s < P.seg.size(); s++)
		run_segment(P, (int)s, blk, gam0, gpos.data(), gcnt, nullptr, nullptr, table, maps, ent_sup, ent_blk, T, &st, base, &tab, WK_MODE_EXACT, 0, scale, nthreads, true);
	out[0] = st.fail ? 0 : gpos[NG] + 2ull * T[G];
	out[1] = st.fail;
	out[2] = st.namb;
	out[3] = P.blk.size();
	out[4] = P.seg.size();
	out[5] = P.table_bytes;
	out[6] = st.sum_d;
	out[7] = (unsigned long long)(1e6 * (st.ngam > 0 ? st.resid2 / st.ngam : 0.0)); /* variance per gamma, in 1e-6 */
	return 0;
}

/* the sequential sampler: T[g] = rejected attempts before group g, total = uniforms consumed */
extern "C" int wk_ref_exact(const int *gam0, int G, const int *gcnt, long s1, long s2, long s3, unsigned long long *T, unsigned long long *total)
{
	isg_cursor c;
	c.s.s1 = (uint32_t)(s1 % ISG_M1); c.s.s2 = (uint32_t)(s2 % ISG_M2); c.s.s3 = (uint32_t)(s3 % ISG_M3);
	c.used = 0;
	c.tape = nullptr;
	unsigned long long used = 0, minimal = 0;
	for (int g = 0; g < G; g++) {
		T[g] = (used - minimal) / 2;
		for (int k = gam0[g]; k < gam0[g + 1]; k++) {
			const uint32_t u0 = c.used;
			(void)isg_rgamma(&c, (double)gcnt[k] + 1.0);
			used += c.used - u0;
			minimal += gcnt[k] == 0 ? 1 : 2;
			if (c.used > (1u << 30)) c.used = 0;
		}
	}
	T[G] = (used - minimal) / 2;
	*total = used;
	return 0;
}

/* interval mode: tables from shape intervals, a lenient walk; returns T and the table bytes along the path */
extern "C" int wk_emul_interval(const int *gam0, int G, const unsigned long long *gpos, const float *alo, const float *ahi, long s1, long s2, long s3, double rho_hi, double sigma,
				double kwin, int seg_groups, float scale, int nthreads, int strict, unsigned long long *T, unsigned char *path_bytes, unsigned long long *out)
{
	WkPlan P;
	if (!wk_plan_build(P, gam0, G, rho_hi, sigma, kwin, seg_groups, false, WK_MODE_INTERVAL)) return 1;
	isg_wh_tables tab;
	isg_wh_tables_init(&tab);
	isg_wh base;
	base.s1 = (uint32_t)(s1 % ISG_M1); base.s2 = (uint32_t)(s2 % ISG_M2); base.s3 = (uint32_t)(s3 % ISG_M3);
	std::vector<WkBlock> blk = P.blk;
	std::vector<unsigned char> table(P.table_bytes + 64);
	std::vector<unsigned short> maps(P.maps_elems + 64);
	std::vector<int> ent_sup(P.sup.size()), ent_blk(P.blk.size());
	WkState st;
	memset(&st, 0, sizeof(st));
	for (size_t s = 0; s < P.seg.size(); s++)
		run_segment(P, (int)s, blk, gam0, gpos, nullptr, alo, ahi, table, maps, ent_sup, ent_blk, T, &st, base, &tab, WK_MODE_INTERVAL, strict, scale, nthreads, true);
	if (!st.fail && path_bytes) {
		for (size_t s = 0; s < P.seg.size(); s++)
			for (int b = P.seg[s].b0; b < P.seg[s].b0 + P.seg[s].nb; b++)
				for (int r = 0; r < blk[b].ng; r++) {
					const int g = blk[b].g0 + r;
					const long col = (long)(T[g] - st.xin[s]) - blk[b].wlo;
					path_bytes[g] = table[blk[b].toff + (size_t)r * blk[b].W + col];
				}
	}
	out[0] = st.fail;
	out[1] = P.blk.size();
	out[2] = P.seg.size();
	out[3] = P.table_bytes;
	return 0;
}

extern "C" float wk_emul_expected_rej(float a) { return wk_expected_rej(a); }

/* uniforms consumed by rdirich over `n` gammas of the given shapes when it starts `pos` uniforms after the seeds */
extern "C" unsigned wk_ref_consumed(unsigned long long pos, const double *shape, int n, long s1, long s2, long s3)
{
	isg_wh_tables tab;
	isg_wh_tables_init(&tab);
	isg_wh base;
	base.s1 = (uint32_t)(s1 % ISG_M1); base.s2 = (uint32_t)(s2 % ISG_M2); base.s3 = (uint32_t)(s3 % ISG_M3);
	isg_cursor c;
	c.s = isg_wh_jump(&tab, base, pos);
	c.used = 0;
	c.tape = nullptr;
	for (int k = 0; k < n; k++) (void)isg_rgamma(&c, shape[k]);
	return c.used;
}

/* the interval resolver end to end (isg_spec_hip.inc's sequence): tables from intervals, a lenient walk, every uncertain byte within
 * +-band of its trajectory replaced by the sequential sampler's consumption for the TRUE shapes, a strict walk over the same tables.
 * Tref: the sequential sampler's trajectory for the true shapes. */
extern "C" int wk_emul_spec(const int *gam0, int G, const unsigned long long *gpos, const float *alo, const float *ahi, const double *atrue, long s1, long s2, long s3, double sigma,
			    double kwin, int seg_groups, int band, int entry_slack, int nthreads, unsigned long long *T, unsigned long long *Tref, unsigned long long *out)
{
	WkPlan P;
	if (!wk_plan_build(P, gam0, G, 0.8, sigma, kwin, seg_groups, false, WK_MODE_INTERVAL, entry_slack)) return 1;
	isg_wh_tables tab;
	isg_wh_tables_init(&tab);
	isg_wh base;
	base.s1 = (uint32_t)(s1 % ISG_M1); base.s2 = (uint32_t)(s2 % ISG_M2); base.s3 = (uint32_t)(s3 % ISG_M3);
	std::vector<WkBlock> blk = P.blk;
	std::vector<unsigned char> table(P.table_bytes + 64);
	std::vector<unsigned short> maps(P.maps_elems + 64);
	std::vector<int> ent_sup(P.sup.size()), ent_blk(P.blk.size());
	WkState st;
	memset(&st, 0, sizeof(st));
	for (size_t s = 0; s < P.seg.size(); s++)
		run_segment(P, (int)s, blk, gam0, gpos, nullptr, alo, ahi, table, maps, ent_sup, ent_blk, T, &st, base, &tab, WK_MODE_INTERVAL, 0, 1.0f, nthreads, true);
	out[0] = st.fail;
	unsigned long long nprobe = 0;
	if (!st.fail) {
		for (size_t s = 0; s < P.seg.size(); s++)
			for (int b = P.seg[s].b0; b < P.seg[s].b0 + P.seg[s].nb; b++)
				for (int r = 0; r < blk[b].ng; r++) {
					const int g = blk[b].g0 + r, n = gam0[g + 1] - gam0[g];
					for (int j = -band; j <= band; j++) {
						const long long x = (long long)T[g] + j, col = x - (long long)st.xin[s] - blk[b].wlo;
						if (x < 0 || col < 0 || col >= blk[b].W) continue;
						unsigned char &v = table[blk[b].toff + (size_t)r * blk[b].W + col];
						if (v == WK_IRR || !(v & WK_UFLAG)) continue;
						const unsigned used = wk_ref_consumed(gpos[gam0[g]] + 2ull * (unsigned long long)x, atrue + gam0[g], n, s1, s2, s3);
						const unsigned c = (used - 2u * (unsigned)n) / 2u;
						v = c <= 126u ? (unsigned char)c : (unsigned char)WK_IRR;
						nprobe++;
					}
				}
		memset((char *)&st + offsetof(WkState, sum_d), 0, sizeof(WkState) - offsetof(WkState, sum_d));
		for (size_t s = 0; s < P.seg.size(); s++)
			run_segment(P, (int)s, blk, gam0, gpos, nullptr, alo, ahi, table, maps, ent_sup, ent_blk, T, &st, base, &tab, WK_MODE_INTERVAL, 1, 1.0f, nthreads, false, 1);
	}
	out[1] = st.fail;
	out[2] = nprobe;
	out[3] = P.seg.size();
	/* reference */
	unsigned long long x = 0;
	for (int g = 0; g < G; g++) {
		const int n = gam0[g + 1] - gam0[g];
		Tref[g] = x;
		x += (wk_ref_consumed(gpos[gam0[g]] + 2ull * x, atrue + gam0[g], n, s1, s2, s3) - 2u * (unsigned)n) / 2u;
	}
	Tref[G] = x;
	return 0;
}

/* the same in two steps, the probes answered by the caller (real update_ZQ: the consumption depends on the candidate's own Z draws) */
static struct {
	WkPlan P; std::vector<WkBlock> blk; std::vector<unsigned char> table; std::vector<unsigned short> maps; std::vector<int> ent_sup, ent_blk; WkState st;
	std::vector<unsigned long long> addr;
} g_spec;
extern "C" long wk_emul_spec_begin(const int *gam0, int G, const unsigned long long *gpos, const float *alo, const float *ahi, long s1, long s2, long s3, double sigma, double kwin,
				   int seg_groups, int band, int entry_slack, int nthreads, unsigned long long *T, int *probe_g, unsigned long long *probe_x, long cap, unsigned long long *out)
{
	WkPlan &P = g_spec.P;
	if (!wk_plan_build(P, gam0, G, 0.8, sigma, kwin, seg_groups, false, WK_MODE_INTERVAL, entry_slack)) return -1;
	isg_wh_tables tab;
	isg_wh_tables_init(&tab);
	isg_wh base;
	base.s1 = (uint32_t)(s1 % ISG_M1); base.s2 = (uint32_t)(s2 % ISG_M2); base.s3 = (uint32_t)(s3 % ISG_M3);
	g_spec.blk = P.blk;
	g_spec.table.assign(P.table_bytes + 64, 0);
	g_spec.maps.assign(P.maps_elems + 64, 0);
	g_spec.ent_sup.assign(P.sup.size(), 0);
	g_spec.ent_blk.assign(P.blk.size(), 0);
	memset(&g_spec.st, 0, sizeof(WkState));
	for (size_t s = 0; s < P.seg.size(); s++)
		run_segment(P, (int)s, g_spec.blk, gam0, gpos, nullptr, alo, ahi, g_spec.table, g_spec.maps, g_spec.ent_sup, g_spec.ent_blk, T, &g_spec.st, base, &tab, WK_MODE_INTERVAL, 0, 1.0f, nthreads, true);
	out[0] = g_spec.st.fail;
	out[1] = g_spec.st.nfail_block;
	g_spec.addr.clear();
	long n = 0;
	if (!g_spec.st.fail)
		for (size_t s = 0; s < P.seg.size(); s++)
			for (int b = P.seg[s].b0; b < P.seg[s].b0 + P.seg[s].nb; b++)
				for (int r = 0; r < g_spec.blk[b].ng; r++)
					for (int j = -band; j <= band; j++) {
						const WkBlock &B = g_spec.blk[b];
						const int g = B.g0 + r;
						const long long x = (long long)T[g] + j, col = x - (long long)g_spec.st.xin[s] - B.wlo;
						if (x < 0 || col < 0 || col >= B.W) continue;
						const unsigned long long a = B.toff + (size_t)r * B.W + col;
						const unsigned char v = g_spec.table[a];
						if (v == WK_IRR || !(v & WK_UFLAG)) continue;
						if (n < cap) { probe_g[n] = g; probe_x[n] = (unsigned long long)x; g_spec.addr.push_back(a); }
						n++;
					}
	return n;
}
extern "C" int wk_emul_spec_finish(const int *gam0, const unsigned long long *gpos, const float *alo, const float *ahi, long s1, long s2, long s3, const unsigned char *probe_c, long n, int nthreads,
				   unsigned long long *T, unsigned long long *out)
{
	WkPlan &P = g_spec.P;
	isg_wh_tables tab;
	isg_wh_tables_init(&tab);
	isg_wh base;
	base.s1 = (uint32_t)(s1 % ISG_M1); base.s2 = (uint32_t)(s2 % ISG_M2); base.s3 = (uint32_t)(s3 % ISG_M3);
	for (long k = 0; k < n && k < (long)g_spec.addr.size(); k++) g_spec.table[g_spec.addr[k]] = probe_c[k];
	memset((char *)&g_spec.st + offsetof(WkState, sum_d), 0, sizeof(WkState) - offsetof(WkState, sum_d));
	for (size_t s = 0; s < P.seg.size(); s++)
		run_segment(P, (int)s, g_spec.blk, gam0, gpos, nullptr, alo, ahi, g_spec.table, g_spec.maps, g_spec.ent_sup, g_spec.ent_blk, T, &g_spec.st, base, &tab, WK_MODE_INTERVAL, 1, 1.0f, nthreads, false, 1);
	out[0] = g_spec.st.fail;
	out[1] = g_spec.st.nfail_block;
	for (size_t s = 0; s <= P.seg.size() && s < 6; s++) out[2 + s] = (unsigned long long)g_spec.st.ent[s];
	return 0;
}

/* one more round on the tables of wk_emul_spec_begin: the given probes patched in, a lenient walk that keeps the tables' origins, and the
 * uncertain bytes that walk's path still holds (band = 0: on the path only).  Returns how many. */
extern "C" long wk_emul_spec_round(const int *gam0, const unsigned long long *gpos, const float *alo, const float *ahi, long s1, long s2, long s3, const unsigned char *probe_c, long n, int band,
				   int nthreads, unsigned long long *T, int *probe_g, unsigned long long *probe_x, long cap, unsigned long long *out)
{
	WkPlan &P = g_spec.P;
	isg_wh_tables tab;
	isg_wh_tables_init(&tab);
	isg_wh base;
	base.s1 = (uint32_t)(s1 % ISG_M1); base.s2 = (uint32_t)(s2 % ISG_M2); base.s3 = (uint32_t)(s3 % ISG_M3);
	for (long k = 0; k < n && k < (long)g_spec.addr.size(); k++) g_spec.table[g_spec.addr[k]] = probe_c[k];
	memset((char *)&g_spec.st + offsetof(WkState, sum_d), 0, sizeof(WkState) - offsetof(WkState, sum_d));
	for (size_t s = 0; s < P.seg.size(); s++)
		run_segment(P, (int)s, g_spec.blk, gam0, gpos, nullptr, alo, ahi, g_spec.table, g_spec.maps, g_spec.ent_sup, g_spec.ent_blk, T, &g_spec.st, base, &tab, WK_MODE_INTERVAL, 0, 1.0f, nthreads, false, 1);
	out[0] = g_spec.st.fail;
	g_spec.addr.clear();
	long m = 0;
	if (!g_spec.st.fail)
		for (size_t s = 0; s < P.seg.size(); s++)
			for (int b = P.seg[s].b0; b < P.seg[s].b0 + P.seg[s].nb; b++)
				for (int r = 0; r < g_spec.blk[b].ng; r++)
					for (int j = -band; j <= band; j++) {
						const WkBlock &B = g_spec.blk[b];
						const int g = B.g0 + r;
						const long long x = (long long)T[g] + j, col = x - (long long)g_spec.st.xin[s] - B.wlo;
						if (x < 0 || col < 0 || col >= B.W) continue;
						const unsigned long long a = B.toff + (size_t)r * B.W + col;
						const unsigned char v = g_spec.table[a];
						if (v == WK_IRR || !(v & WK_UFLAG)) continue;
						if (m < cap) { probe_g[m] = g; probe_x[m] = (unsigned long long)x; g_spec.addr.push_back(a); }
						m++;
					}
	return m;
}
