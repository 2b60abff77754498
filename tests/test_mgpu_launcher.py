"""The multi-GPU launcher (instruct_amd/host/instruct_mgpu.c): chains (-c) or the K values of a K scan sharded one worker
process per GPU around the UNMODIFIED driver program.

CPU part: the launcher around the pure reference binary (oracle/_ref/InStruct_ref, development container only) with two
worker processes -- its combined result file holds exactly the chain blocks of two separate `-c 1 -s ...` reference runs
(tests/golden/mgpu_rank*_cli_output.txt) in rank order, and the Gelman-Rubin value of their concatenated samples.
GPU part: the same through the MI355X drop-in (samples exchanged as raw doubles / by ncclAllGather)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import golden_util as gu
import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MGPU = os.path.join(ROOT, "instruct_amd", "host", "instruct_mgpu")
MG = gu.make_golden


def _build():
    from instruct_amd import build
    build.build_host()
    orc.build()


def _blocks(path):
    """chain_stat blocks of a result file (bytes), titles normalised to b'Chain#N'"""
    data = open(path, "rb").read()
    parts = data.split(b"\n\n\nChain#")
    out = []
    for p in parts[1:]:
        body = p.split(b"There is only one MCMC.")[0].split(b"\n\nThe Gelman-Rubin statistics")[0]
        out.append(re.sub(rb"^\d+", b"N", body))
    return parts[0], out


def _cf_values(path):
    return [float(x) for x in open(path).read().split("\n")[1].split()]


def _run_launcher(exe, tmp_path, chains, extra=(), gather="file", gpus=1):
    out = tmp_path / "comb.txt"
    cmd = [MGPU, "--exe", exe, "--gpus", str(gpus), "--gather", gather, "--", "-d", os.path.join(gu.GOLDEN, "c1.txt"), "-o", str(out)] + \
        MG.MGPU_BASE + ["-c", str(chains), "-g", "1", "-s"] + [str(s) for s in MG.MGPU_SEEDS] + list(extra)
    log = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert log.returncode == 0 and b"THE JOB IS SUCCESSFULLY FINISHED" in log.stdout, log.stdout[-3000:]
    return out, log.stdout


def _check_combined(out, stdout, exact_samples):
    head, blocks = _blocks(str(out))
    assert b"Chain Number=2" in head and b"instruct_mgpu" in head
    want = [_blocks(os.path.join(gu.GOLDEN, "mgpu_rank%d_cli_output.txt" % r))[1][0] for r in range(2)]
    assert blocks == want     # each rank's chain_stat block == the separate reference run's, in rank order
    titles = re.findall(rb"\n\n\nChain#(\d+)", open(str(out), "rb").read())
    assert titles == [b"1", b"2"]
    samples = np.array(_cf_values(os.path.join(gu.GOLDEN, "mgpu_rank0_cf.txt")) + _cf_values(os.path.join(gu.GOLDEN, "mgpu_rank1_cf.txt")))
    gr = orc.gelman_rubin(samples, 2, 6)
    m = re.search(rb"The Gelman-Rubin statistics for the convergence of log-likelihood is (\S+)\.\n", open(str(out), "rb").read())
    assert m, "no Gelman-Rubin line"
    got = float(m.group(1))
    # the statistic of the concatenated samples: exactly (to the printed 6 decimals) when the samples came from the
    # reference's own -cf dump, within the dump's 6-decimal rounding when the launcher had the exact doubles
    assert abs(got - gr) <= (2e-4 if exact_samples else 1.1e-6) * max(1.0, abs(gr)), (got, gr)
    assert ("The Gelman-Rubin statistics of log-likelihood is %f" % got).encode() in stdout


def test_launcher_two_processes_around_the_reference_binary(tmp_path):
    exe = os.path.join(ROOT, "oracle", "_ref", "InStruct_ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/InStruct_ref not built (development container only)")
    _build()
    out, stdout = _run_launcher(exe, tmp_path, 2)
    _check_combined(out, stdout, exact_samples=False)


def test_launcher_k_scan_shards_the_values_of_k(tmp_path):
    """-ik 1 -kv 2 3 through the launcher: one worker per K (each running its chains back to back), sections in the
    reference's layout, the optimal K = the K whose best chain has the smallest DIC (InStruct.c:586-591)"""
    exe = os.path.join(ROOT, "oracle", "_ref", "InStruct_ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/InStruct_ref not built (development container only)")
    _build()
    out = tmp_path / "k.txt"
    base = ["-d", os.path.join(gu.GOLDEN, "c1.txt")] + [a for a in MG.KSCAN_CLI]
    log = subprocess.run([MGPU, "--exe", exe, "--gpus", "1", "--", "-o", str(out)] + base, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert log.returncode == 0, log.stdout[-3000:]
    data = open(str(out), "rb").read()
    assert data.count(b"The current K is 2\n") == 1 and data.count(b"The current K is 3\n") == 1
    assert data.index(b"The current K is 2") < data.index(b"The current K is 3")
    # every K section is what a separate run with that K and the worker's seeds writes
    for idx, K in enumerate((2, 3)):
        sep = tmp_path / ("sep%d.txt" % K)
        cli = [a for a in MG.KSCAN_CLI]
        for flag, n in (("-ik", 1), ("-kv", 2), ("-K", 1), ("-s", 3)):
            k = cli.index(flag)
            del cli[k:k + n + 1]
        cmd = [exe, "-d", os.path.join(gu.GOLDEN, "c1.txt"), "-o", str(sep)] + cli + ["-K", str(K), "-ik", "0", "-s"] + [str(s + idx) for s in (13, 4, 1972)]
        assert subprocess.run(cmd, stdout=subprocess.DEVNULL, timeout=600).returncode == 0
        sec = data.split(b"The current K is %d\n" % K)[1].split(b"\n\nThe current K is")[0].split(b"\n\nThe range of value for K")[0]
        assert sec == b"\n\n\nChain#" + open(str(sep), "rb").read().split(b"\n\n\nChain#", 1)[1]
    dic = {}
    for K in (2, 3):
        sec = data.split(b"The current K is %d\n" % K)[1].split(b"The current K is")[0]
        dic[K] = min(float(x) for x in re.findall(rb"information criterion of this model is (-?[\d.]+)\.\n", sec))
    best = min(dic, key=dic.get)
    assert data.endswith(b"\n\nThe range of value for K is (2 - 3)!\nThe optimal K is %d\n" % best)


@pytest.mark.gpu
def test_launcher_around_the_drop_in_equals_separate_reference_runs(tmp_path):
    exe = os.path.join(ROOT, "oracle", "_ref", "InStruct_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/InStruct_hip not built (needs the reference objects; built in the dev container)")
    _build()
    out, stdout = _run_launcher(exe, tmp_path, 2)      # two chains sharing the one GPU of the test box: samples through files
    _check_combined(out, stdout, exact_samples=True)


@pytest.mark.gpu
def test_launcher_rccl_gather_one_rank(tmp_path):
    """--gather rccl with one chain on one GPU: the drop-in loads librccl, creates the communicator from the id file,
    runs ncclAllGather on device buffers and leaves the gathered vector; the block equals the reference run's"""
    exe = os.path.join(ROOT, "oracle", "_ref", "InStruct_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/InStruct_hip not built (needs the reference objects; built in the dev container)")
    _build()
    out = tmp_path / "one.txt"
    cmd = [MGPU, "--exe", exe, "--gpus", "1", "--gather", "rccl", "--keep", "--", "-d", os.path.join(gu.GOLDEN, "c1.txt"), "-o", str(out)] + \
        MG.mgpu_rank_cli(0)
    log = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert log.returncode == 0, log.stdout[-3000:]
    d = str(out) + ".mgpu"
    allv = np.fromfile(os.path.join(d, "convg_all.bin"))
    mine = np.fromfile(os.path.join(d, "convg.0.bin"))
    assert allv.size == 6 and np.array_equal(allv, mine)
    assert np.allclose(mine, _cf_values(os.path.join(gu.GOLDEN, "mgpu_rank0_cf.txt")), rtol=0, atol=1e-6)
    assert _blocks(str(out))[1] == _blocks(os.path.join(gu.GOLDEN, "mgpu_rank0_cli_output.txt"))[1]
    assert open(str(out), "rb").read().endswith(b"There is only one MCMC. No need to check the convergence.\n")


@pytest.mark.gpu
def test_abi_rccl_all_gather_on_device_buffers(tmp_path):
    from instruct_amd import capi, synth
    geno, an, mi = synth.code_diploid(synth.raw_alleles(20, 40, 3, 2, 2, 0.0, 5))
    h = capi.HipChain(geno, an, mi, 3)
    mine = -3000.0 + np.arange(20, dtype=np.float64) * 0.37
    got = h.gather_convg(0, 1, tmp_path / "id", mine)
    assert np.array_equal(got, mine)
    h.close()


def test_launcher_fails_fast_when_a_worker_dies(tmp_path):
    """A worker that dies (bad flag, HIP error, nrerror) must end the run at once: its siblings may be blocked in the RCCL
    rendezvous waiting for it.  Rank 1 of four exits with an error after a moment, the others would run for a minute: the
    launcher reports rank 1's output, leaves the `abort` file for pollers, stops the others, returns non-zero within seconds
    and leaves no child behind."""
    import signal
    import time
    from instruct_amd import build
    build.build_host()
    exe = tmp_path / "worker.sh"
    pids = tmp_path / "pids"
    exe.write_text("#!/bin/sh\necho $$ >> %s\nif [ \"$INSTRUCT_MGPU_RANK\" = 1 ]; then sleep 0.3; echo 'ERROR: \nworker one is unwell'; exit 3; fi\nexec sleep 60\n" % pids)
    os.chmod(str(exe), 0o755)
    out = tmp_path / "ff.txt"
    t0 = time.time()
    log = subprocess.run([MGPU, "--exe", str(exe), "--gpus", "4", "--gather", "rccl", "--keep", "--", "-o", str(out), "-c", "4", "-K", "2"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60)
    dt = time.time() - t0
    assert log.returncode != 0 and dt < 10, (log.returncode, dt)
    assert b"worker 1 failed" in log.stdout and b"worker one is unwell" in log.stdout
    assert os.path.exists(str(out) + ".mgpu/abort")
    time.sleep(0.2)
    for pid in [int(x) for x in open(str(pids)).read().split()]:
        try:
            os.kill(pid, 0)
            alive = True
        except ProcessLookupError:
            alive = False
        assert not alive, "worker %d still running" % pid


@pytest.mark.gpu
def test_launcher_eight_chains_on_one_gpu_through_files(tmp_path):
    """More workers than GPUs (`--gpus 1 --gather file`, W worker processes sharing the test box's one MI355X): every chain's block equals
    the separate `-c 1 -s s+r` run of the same program (ranks 0 and 1 also the reference's own, tests/golden/mgpu_rank*), the
    Gelman-Rubin line is the statistic of the eight chains' exact samples."""
    exe = os.path.join(ROOT, "oracle", "_ref", "InStruct_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/InStruct_hip not built (needs the reference objects; built in the dev container)")
    _build()
    W = 4   # (the test box lets at most 6 processes use the GPU together, and this test runner -- which has run GPU tests before -- is one of them)
    out = tmp_path / "eight.txt"
    cmd = [MGPU, "--exe", exe, "--gpus", "1", "--gather", "file", "--keep", "--", "-d", os.path.join(gu.GOLDEN, "c1.txt"), "-o", str(out)] + \
        MG.MGPU_BASE + ["-c", str(W), "-g", "1", "-s"] + [str(s) for s in MG.MGPU_SEEDS]
    log = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert log.returncode == 0 and b"THE JOB IS SUCCESSFULLY FINISHED" in log.stdout, log.stdout[-3000:]
    head, blocks = _blocks(str(out))
    assert len(blocks) == W and ("Chain Number=%d" % W).encode() in head
    for r in range(2):
        assert blocks[r] == _blocks(os.path.join(gu.GOLDEN, "mgpu_rank%d_cli_output.txt" % r))[1][0]
    samples = []
    for r in range(W):
        sep = tmp_path / ("sep%d.txt" % r)
        cf = tmp_path / ("sep%d.cf" % r)
        one = subprocess.run([exe, "-d", os.path.join(gu.GOLDEN, "c1.txt"), "-o", str(sep), "-cf", str(cf)] + MG.mgpu_rank_cli(r), stdout=subprocess.DEVNULL, timeout=600)
        assert one.returncode == 0
        assert blocks[r] == _blocks(str(sep))[1][0], r
        mine = np.fromfile(os.path.join(str(out) + ".mgpu", "convg.%d.bin" % r))
        assert mine.size == 6 and np.allclose(mine, _cf_values(str(cf)), rtol=0, atol=1e-6)
        samples.append(mine)
    gr = orc.gelman_rubin(np.concatenate(samples), W, 6)
    m = re.search(rb"The Gelman-Rubin statistics for the convergence of log-likelihood is (\S+)\.\n", open(str(out), "rb").read())
    # (W chains of six samples: repperchain = 6 / W = 1 and the reference's formula divides by rep - 1 = 0, check_converg.c:121-141: nan there, nan here)
    got = float(m.group(1))
    assert m and ((np.isnan(got) and np.isnan(gr)) or abs(got - gr) <= 1.1e-6 * max(1.0, abs(gr))), (got, gr)


@pytest.mark.gpu
def test_launcher_k_scan_around_the_drop_in(tmp_path):
    """-ik 1 -kv 2 4 sharded one worker per K around the drop-in (three processes on the one GPU): every section is what a separate
    run of the same program with that K and the worker's seeds writes, the optimal K the one with the smallest DIC"""
    exe = os.path.join(ROOT, "oracle", "_ref", "InStruct_hip")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/InStruct_hip not built (needs the reference objects; built in the dev container)")
    _build()
    out = tmp_path / "k.txt"
    cli = [a for a in MG.KSCAN_CLI]
    k = cli.index("-kv")
    cli[k + 1:k + 3] = ["2", "4"]
    log = subprocess.run([MGPU, "--exe", exe, "--gpus", "1", "--", "-d", os.path.join(gu.GOLDEN, "c1.txt"), "-o", str(out)] + cli, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert log.returncode == 0, log.stdout[-3000:]
    data = open(str(out), "rb").read()
    dic = {}
    for idx, K in enumerate((2, 3, 4)):
        assert data.count(b"The current K is %d\n" % K) == 1
        sep = tmp_path / ("sep%d.txt" % K)
        one = [a for a in MG.KSCAN_CLI]
        for flag, n in (("-ik", 1), ("-kv", 2), ("-K", 1), ("-s", 3)):
            q = one.index(flag)
            del one[q:q + n + 1]
        cmd = [exe, "-d", os.path.join(gu.GOLDEN, "c1.txt"), "-o", str(sep)] + one + ["-K", str(K), "-ik", "0", "-s"] + [str(s + idx) for s in (13, 4, 1972)]
        assert subprocess.run(cmd, stdout=subprocess.DEVNULL, timeout=600).returncode == 0
        sec = data.split(b"The current K is %d\n" % K)[1].split(b"\n\nThe current K is")[0].split(b"\n\nThe range of value for K")[0]
        assert sec == b"\n\n\nChain#" + open(str(sep), "rb").read().split(b"\n\n\nChain#", 1)[1], K
        dic[K] = min(float(x) for x in re.findall(rb"information criterion of this model is (-?[\d.]+)\.\n", sec))
    assert data.endswith(b"\n\nThe range of value for K is (2 - 4)!\nThe optimal K is %d\n" % min(dic, key=dic.get))
