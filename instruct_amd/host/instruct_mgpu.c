/*
 * instruct_mgpu.c -- one node, several MI355X: the chains (-c) of an InStruct run, or the values of K of a K scan
 * (-ik 1 -kv a b), sharded one per GPU.
 *
 * The reference runs its chains back to back in one process on one global random stream (InStruct.c:182-193) and the K
 * scan one K after the other (inf_K_val, InStruct.c:536-601).  Both are embarrassingly parallel.  This launcher -- plain
 * C, no GPU call of its own, so it can start the workers before anything touches a device -- runs the UNMODIFIED driver
 * program (linked with the MI355X drop-in sampler) once per chain / per K:
 *
 *   chains:  rank r = `<exe> <your flags> -c 1 -g 1 -s s1+r s2+r s3+r -o <dir>/out.<r>` on GPU r mod G.  Each rank's
 *            stored log-likelihood samples (CONVG.convg_ld, mcmc.c:223-224) are exchanged with ONE ncclAllGather over
 *            RCCL / xGMI inside the drop-in (isg_gather_convg; --gather file: through <dir>/convg.<r>.bin instead, for
 *            more chains than GPUs).  The launcher evaluates GelmanRubin (check_converg.c:100-153, with its
 *            repperchain = ckrep / chains indexing, :121-137) and writes ONE result file: rank 0's header, every chain's
 *            chain_stat block (result_analysis.c:34-70) in rank order titled Chain#<r+1>, then the reference's
 *            Gelman-Rubin line (check_converg.c:69).  Not equal to one reference `-c C` run (there the chains share the
 *            stream); equal to C separate `-c 1 -s ...` runs plus the statistic of their samples.
 *   K scan:  one worker per K = `<exe> <your flags> -ik 0 -K k -s s1+i s2+i s3+i -o <dir>/out.<i>`, round-robin over the
 *            GPUs; the launcher reads each chain's DIC (result_analysis.c:389-412), picks the K whose best chain has the
 *            smallest DIC (InStruct.c:586-591) and writes the reference's layout: "The current K is k" sections, then
 *            the range and the optimal K (InStruct.c:553, 595-597).
 *
 * usage: instruct_mgpu [--exe PATH] [--gpus G] [--gather rccl|file] [--keep] -- <InStruct flags ...>
 *        (--exe defaults to $INSTRUCT_EXE or ./InStruct; -o is required among the flags)
 */
#define _GNU_SOURCE
#include <errno.h>
#include <math.h>
#include <signal.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#define MAXW 256

static void die(const char *msg, const char *arg)
{
	fprintf(stdout, "ERROR: \n%s%s\n...now exiting to system...\n", msg, arg ? arg : ""); /* nrerror's convention, nrutil.c:9-16 */
	exit(1);
}

/* GelmanRubin, check_converg.c:100-153, as the reference evaluates it (per-chain length totrep / numchains) */
static double gelman_rubin(const double *vec, int numchains, int totrep)
{
	const int rep = totrep / numchains;
	double psi = 0, W = 0, B = 0, V;
	double *psii = (double *)calloc((size_t)numchains, sizeof(double)), *S = (double *)calloc((size_t)numchains, sizeof(double));
	int i, j;
	for (i = 0; i < numchains; i++) {
		for (j = 0; j < rep; j++) psii[i] += vec[i * rep + j];
		psii[i] = psii[i] / rep;
		psi = psi + psii[i];
	}
	psi = psi / numchains;
	for (i = 0; i < numchains; i++) {
		for (j = 0; j < rep; j++) S[i] += (vec[i * rep + j] - psii[i]) * (vec[i * rep + j] - psii[i]);
		S[i] = S[i] / (rep - 1);
		W += S[i];
	}
	W = W / numchains;
	for (i = 0; i < numchains; i++) B += (psii[i] - psi) * (psii[i] - psi);
	B = (B * rep) / (numchains - 1);
	V = (W * (rep - 1)) / rep + B / rep;
	free(psii);
	free(S);
	return V / W;
}

static char *slurp(const char *path, size_t *len)
{
	FILE *f = fopen(path, "rb");
	char *buf;
	long n;
	if (!f) return NULL;
	fseek(f, 0, SEEK_END);
	n = ftell(f);
	fseek(f, 0, SEEK_SET);
	buf = (char *)malloc((size_t)n + 1);
	if (!buf || fread(buf, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(buf); return NULL; }
	fclose(f);
	buf[n] = 0;
	*len = (size_t)n;
	return buf;
}

/* memmem on buffers that hold NUL bytes (the result file has one per chain title, result_analysis.c:395-396) */
static char *find(char *hay, size_t n, const char *needle)
{
	return (char *)memmem(hay, n, needle, strlen(needle));
}

/* number of values a flag of the driver takes (InStruct.c:249-433: every flag takes one, -s three, -kv two) */
static int arity(const char *flag)
{
	if (strcmp(flag, "-s") == 0) return 3;
	if (strcmp(flag, "-kv") == 0) return 2;
	return 1;
}

typedef struct {
	pid_t pid;
	int status;
	char out[4096], log[4096], cf[4096];
} worker;

static pid_t spawn(const char *exe, char **argv, const char *logpath, int device, int rank, int world, const char *dir, const char *gather, int shared_gpu)
{
	pid_t pid = fork();
	if (pid < 0) die("fork failed", NULL);
	if (pid == 0) {
		char buf[64];
		if (freopen(logpath, "w", stdout) == NULL) _exit(127);
		dup2(fileno(stdout), fileno(stderr));
		snprintf(buf, sizeof(buf), "%d", device);
		setenv("INSTRUCT_DEVICE", buf, 1);
		/* more workers than GPUs: the replay update_ZQ resolver launches block by block (its one-launch form wants the whole chip
		 * to itself; processes cannot see each other's contexts) -- unless the caller has set the switch */
		if (shared_gpu) setenv("INSTRUCT_ZQ_RESOLVE_PERSIST", "0", 0);
		if (world > 0) {
			snprintf(buf, sizeof(buf), "%d", rank);
			setenv("INSTRUCT_MGPU_RANK", buf, 1);
			snprintf(buf, sizeof(buf), "%d", world);
			setenv("INSTRUCT_MGPU_WORLD", buf, 1);
			setenv("INSTRUCT_MGPU_DIR", dir, 1);
			setenv("INSTRUCT_MGPU_GATHER", gather, 1);
		}
		execv(exe, argv);
		_exit(127);
	}
	return pid;
}

int main(int argc, char **argv)
{
	const char *exe = getenv("INSTRUCT_EXE"), *gather = NULL, *outfile = NULL, *cffile = NULL;
	int gpus = 0, keep = 0, first = 1, i, j, chains = 2, ckrep = 20, inf_K = 0, kv0 = 1, kv1 = 0, nflags, W, r;
	long seeds[3] = {13, 4, 1972}; /* random.c:10-12 */
	char dir[3900], **flags;
	worker *w;
	if (!exe) exe = "./InStruct";
	for (i = 1; i < argc; i++) {
		if (strcmp(argv[i], "--") == 0) { first = i + 1; break; }
		if (strcmp(argv[i], "--exe") == 0 && i + 1 < argc) exe = argv[++i];
		else if (strcmp(argv[i], "--gpus") == 0 && i + 1 < argc) gpus = atoi(argv[++i]);
		else if (strcmp(argv[i], "--gather") == 0 && i + 1 < argc) gather = argv[++i];
		else if (strcmp(argv[i], "--keep") == 0) keep = 1;
		else { first = i; break; }
		first = i + 1;
	}
	flags = argv + first;
	nflags = argc - first;
	for (i = 0; i < nflags; i++) {
		const char *f = flags[i];
		const int a = arity(f);
		if (f[0] != '-') continue;
		if (i + a >= nflags) die("instruct_mgpu: missing value after ", f);
		if (strcmp(f, "-o") == 0) outfile = flags[i + 1];
		if (strcmp(f, "-cf") == 0) cffile = flags[i + 1];
		if (strcmp(f, "-c") == 0) chains = atoi(flags[i + 1]);
		if (strcmp(f, "-r") == 0) ckrep = atoi(flags[i + 1]);
		if (strcmp(f, "-ik") == 0) inf_K = atoi(flags[i + 1]);
		if (strcmp(f, "-kv") == 0) { kv0 = atoi(flags[i + 1]); kv1 = atoi(flags[i + 2]); }
		if (strcmp(f, "-s") == 0) for (j = 0; j < 3; j++) seeds[j] = atol(flags[i + 1 + j]);
		i += a;
	}
	if (!outfile) die("instruct_mgpu: -o output_file is required", NULL);
	if (inf_K == 1 && (kv0 < 1 || kv1 < kv0)) die("instruct_mgpu: the K scan needs -kv n_small n_large with 1 <= n_small <= n_large", NULL);
	W = inf_K == 1 ? kv1 - kv0 + 1 : chains;
	if (W < 1 || W > MAXW) die("instruct_mgpu: between 1 and 256 workers", NULL);
	if (gpus < 1) gpus = W; /* the node this is meant for has one MI355X per worker */
	if (!gather) gather = (inf_K == 0 && W <= gpus && W > 1) ? "rccl" : "file";
	if (strcmp(gather, "rccl") == 0 && W > gpus) die("instruct_mgpu: --gather rccl needs one GPU per chain (RCCL does not put two ranks on one device)", NULL);
	snprintf(dir, sizeof(dir), "%s.mgpu", outfile);
	if (mkdir(dir, 0777) != 0 && errno != EEXIST) die("instruct_mgpu: cannot create the rendezvous directory ", dir);
	{ /* stale files of an earlier run would satisfy the pollers */
		char p[4200];
		snprintf(p, sizeof(p), "%s/nccl_id", dir); unlink(p);
		snprintf(p, sizeof(p), "%s/abort", dir); unlink(p);
		snprintf(p, sizeof(p), "%s/convg_all.bin", dir); unlink(p);
		for (r = 0; r < W; r++) { snprintf(p, sizeof(p), "%s/convg.%d.bin", dir, r); unlink(p); }
	}
	w = (worker *)calloc((size_t)W, sizeof(worker));
	/* ---- start the workers: the caller's flags with -c / -g / -s / -o / -cf / -ik / -kv / -K replaced ---- */
	for (r = 0; r < W; r++) {
		char **av = (char **)calloc((size_t)nflags + 32, sizeof(char *));
		char sbuf[3][32], kbuf[32];
		int n = 0;
		av[n++] = (char *)exe;
		for (i = 0; i < nflags; i++) {
			const char *f = flags[i];
			const int a = (f[0] == '-') ? arity(f) : 0;
			const int drop = strcmp(f, "-c") == 0 || strcmp(f, "-g") == 0 || strcmp(f, "-s") == 0 || strcmp(f, "-o") == 0 || strcmp(f, "-cf") == 0 ||
					 strcmp(f, "-ik") == 0 || strcmp(f, "-kv") == 0 || (inf_K == 1 && strcmp(f, "-K") == 0);
			if (!drop) for (j = 0; j <= a && i + j < nflags; j++) av[n++] = flags[i + j];
			i += a;
		}
		snprintf(w[r].out, sizeof(w[r].out), "%s/out.%d", dir, r);
		snprintf(w[r].log, sizeof(w[r].log), "%s/log.%d", dir, r);
		unlink(w[r].out); /* the driver appends (fopen "a+", result_analysis.c:45) */
		av[n++] = "-o"; av[n++] = w[r].out;
		av[n++] = "-g"; av[n++] = "1"; /* CONVG is only allocated with -g 1 (InStruct.c:181) and the sampler always writes into it */
		av[n++] = "-s";
		for (j = 0; j < 3; j++) { snprintf(sbuf[j], sizeof(sbuf[j]), "%ld", seeds[j] + r); av[n++] = sbuf[j]; }
		if (inf_K == 1) {
			char cbuf[32];
			snprintf(kbuf, sizeof(kbuf), "%d", kv0 + r);
			snprintf(cbuf, sizeof(cbuf), "%d", chains);
			av[n++] = "-K"; av[n++] = kbuf;
			av[n++] = "-c"; av[n++] = strdup(cbuf);
			av[n++] = "-ik"; av[n++] = "0";
		} else {
			av[n++] = "-c"; av[n++] = "1";
			snprintf(w[r].cf, sizeof(w[r].cf), "%s/cf.%d", dir, r); /* the samples as text too (check_converg.c:75-89) */
			av[n++] = "-cf"; av[n++] = w[r].cf;
		}
		av[n] = NULL;
		w[r].pid = spawn(exe, av, w[r].log, r % gpus, r, inf_K == 1 ? 0 : W, dir, gather, W > gpus);
		free(av);
	}
	/* ---- wait for them in whatever order they finish.  The first worker that fails ends the run: its siblings may be waiting for it
	 * (the RCCL rendezvous, ncclCommInitRank, the all-gather) and would wait for ever, so an `abort` file tells the pollers, the rest are
	 * terminated, and the failing worker's output is shown.  Nothing is restarted: a failed rank is a failed run. ---- */
	{
		int left = W, failed = -1;
		while (left > 0) {
			int st = 0;
			const pid_t pid = waitpid(-1, &st, 0);
			if (pid < 0) {
				if (errno == EINTR) continue;
				die("waitpid failed", NULL);
			}
			for (r = 0; r < W && w[r].pid != pid; r++) { }
			if (r == W) continue; /* not one of ours */
			w[r].status = st;
			w[r].pid = 0;
			left--;
			if (failed < 0 && (!WIFEXITED(st) || WEXITSTATUS(st) != 0)) {
				char p[4200];
				FILE *a;
				time_t t0;
				failed = r;
				snprintf(p, sizeof(p), "%s/abort", dir);
				if ((a = fopen(p, "w")) != NULL) { fprintf(a, "worker %d failed\n", r); fclose(a); }
				for (j = 0; j < W; j++) if (w[j].pid > 0) kill(w[j].pid, SIGTERM);
				/* a few seconds for them to go, then no more patience */
				t0 = time(NULL);
				while (left > 0) {
					const pid_t q = waitpid(-1, &st, WNOHANG);
					if (q > 0) {
						for (j = 0; j < W && w[j].pid != q; j++) { }
						if (j < W) { w[j].pid = 0; w[j].status = st; left--; }
					} else if (q < 0 && errno != EINTR) {
						break;
					} else {
						if (time(NULL) - t0 > 5) { for (j = 0; j < W; j++) if (w[j].pid > 0) kill(w[j].pid, SIGKILL); }
						usleep(20000);
					}
				}
			}
		}
		if (failed >= 0) {
			size_t n = 0;
			char *log = slurp(w[failed].log, &n);
			fprintf(stdout, "instruct_mgpu: worker %d failed (status %d), the other workers were stopped; its output follows\n%s\n", failed, w[failed].status,
				log ? log + (n > 3000 ? n - 3000 : 0) : "");
			die("a worker failed; see ", w[failed].log);
		}
	}
	/* ---- assemble the result file ---- */
	{
		FILE *o = fopen(outfile, "wb");
		size_t n0 = 0;
		char *f0 = slurp(w[0].out, &n0), *blk;
		const char *title = "\n\n\nChain#";
		if (!o) die("Cannot open output file!", NULL);
		if (!f0 || !(blk = find(f0, n0, title))) die("instruct_mgpu: no chain block in ", w[0].out);
		/* the header is rank 0's (InStruct.c:450-531) with this program's own command line and chain count */
		{
			const char *mark = "Command line arguments:\n", *cn = "    Chain Number=";
			char *p = find(f0, (size_t)(blk - f0), mark), *q, *e;
			if (p) {
				p += strlen(mark);
				fwrite(f0, 1, (size_t)(p - f0), o);
				fprintf(o, "    ");
				for (i = 0; i < argc; i++) fprintf(o, "%s ", argv[i]);
				e = (char *)memchr(p, '\n', (size_t)(blk - p));
				p = e ? e : p;
			} else {
				p = f0;
			}
			q = find(p, (size_t)(blk - p), cn);
			if (q && inf_K == 0) {
				fwrite(p, 1, (size_t)(q - p), o);
				fprintf(o, "%s%d", cn, chains);
				e = (char *)memchr(q, '\n', (size_t)(blk - q));
				p = e ? e : q;
			}
			fwrite(p, 1, (size_t)(blk - p), o);
		}
		if (inf_K == 0) {
			double *all = (double *)malloc(sizeof(double) * (size_t)ckrep * W);
			int have = 0;
			for (r = 0; r < W; r++) {
				size_t n = 0;
				char *f = r ? slurp(w[r].out, &n) : f0, *b, *e, *t;
				if (!r) n = n0;
				if (!f || !(b = find(f, n, title))) die("instruct_mgpu: no chain block in ", w[r].out);
				e = find(b, n - (size_t)(b - f), "There is only one MCMC.");
				if (!e) e = f + n;
				/* "\n\n\nChain#1" -> "\n\n\nChain#<r+1>" (every worker ran its chain as chain 1) */
				t = b + strlen(title);
				fwrite(b, 1, strlen(title), o);
				fprintf(o, "%d", r + 1);
				while (t < e && *t >= '0' && *t <= '9') t++;
				fwrite(t, 1, (size_t)(e - t), o);
				if (r) free(f);
			}
			{ /* the samples: the RCCL-gathered vector if the ranks left one, else every rank's own file */
				char p[4200];
				size_t n = 0;
				char *g;
				snprintf(p, sizeof(p), "%s/convg_all.bin", dir);
				g = slurp(p, &n);
				if (g && n == sizeof(double) * (size_t)ckrep * W) { memcpy(all, g, n); have = 1; }
				free(g);
				if (!have) {
					have = 1;
					for (r = 0; r < W; r++) {
						snprintf(p, sizeof(p), "%s/convg.%d.bin", dir, r);
						g = slurp(p, &n);
						if (g && n == sizeof(double) * (size_t)ckrep) memcpy(all + (size_t)r * ckrep, g, n);
						else have = 0;
						free(g);
					}
				}
				if (!have) { /* a driver without the drop-in sampler (the pure reference program): its -cf dump, 6 decimals */
					have = 1;
					for (r = 0; r < W && have; r++) {
						char *q;
						g = slurp(w[r].cf, &n);
						q = g ? strchr(g, '\n') : NULL;
						for (i = 0; i < ckrep && q; i++) {
							char *end;
							all[(size_t)r * ckrep + i] = strtod(q, &end);
							if (end == q) q = NULL; else q = end;
						}
						if (!q) have = 0;
						free(g);
					}
				}
			}
			if (W == 1) {
				fprintf(o, "There is only one MCMC. No need to check the convergence.\n"); /* check_converg.c:61 */
			} else if (have) {
				const double gr = gelman_rubin(all, W, ckrep);
				fprintf(stdout, "The Gelman-Rubin statistics of log-likelihood is %f\n", gr); /* check_converg.c:66-69 */
				fprintf(o, "\n\nThe Gelman-Rubin statistics for the convergence of log-likelihood is %f.\n", gr);
			} else {
				die("instruct_mgpu: the workers left no log-likelihood samples (is the program linked with the MI355X drop-in sampler?)", NULL);
			}
			if (cffile && have) { /* check_converg.c:75-89 */
				FILE *c = fopen(cffile, "w");
				if (!c) die("ERROR: Cannot open convergence file!\n", NULL);
				fprintf(c, "Values of log-likelihood:\n");
				for (i = 0; i < ckrep * W; i++) fprintf(c, i ? " %f " : "%f ", all[i]);
				fprintf(c, "\n");
				fclose(c);
			}
			free(all);
		} else {
			double best = 0;
			int kbest = kv0;
			for (r = 0; r < W; r++) {
				size_t n = 0;
				char *f = r ? slurp(w[r].out, &n) : f0, *b, *p;
				const char *dm = "The Deviance information criterion of this model is ";
				double kmin = 0;
				int seen = 0;
				if (!r) n = n0;
				if (!f || !(b = find(f, n, title))) die("instruct_mgpu: no chain block in ", w[r].out);
				fprintf(o, "\n\nThe current K is %d\n", kv0 + r); /* InStruct.c:553 */
				fwrite(b, 1, n - (size_t)(b - f), o);
				for (p = b; (p = find(p, n - (size_t)(p - f), dm)) != NULL; p += strlen(dm)) {
					const double v = atof(p + strlen(dm));
					if (!seen || v < kmin) kmin = v; /* the chain with the smallest DIC stands for this K (InStruct.c:586-589) */
					seen = 1;
				}
				if (!seen) die("instruct_mgpu: no DIC line in ", w[r].out);
				if (r == 0 || kmin < best) { best = kmin; kbest = kv0 + r; }
				if (r) free(f);
			}
			fprintf(o, "\n\nThe range of value for K is (%d - %d)!\n", kv0, kv1); /* InStruct.c:595-597 */
			fprintf(o, "The optimal K is %d\n", kbest);
		}
		free(f0);
		fclose(o);
	}
	if (!keep) {
		char p[4200];
		for (r = 0; r < W; r++) { unlink(w[r].out); unlink(w[r].log); if (w[r].cf[0]) unlink(w[r].cf); snprintf(p, sizeof(p), "%s/convg.%d.bin", dir, r); unlink(p); }
		snprintf(p, sizeof(p), "%s/nccl_id", dir); unlink(p);
		snprintf(p, sizeof(p), "%s/convg_all.bin", dir); unlink(p);
		rmdir(dir);
	}
	fprintf(stdout, "THE JOB IS SUCCESSFULLY FINISHED\n"); /* InStruct.c:200 */
	return 0;
}
