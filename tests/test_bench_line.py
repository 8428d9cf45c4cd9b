"""The bench line contract, checked on the line committed from the last device run (profiles/r6_bench_default.json):
the keys the driver parses, BASELINE.json's metric and headline workload, a roofline object that follows from its own
inputs, a CPU baseline with its sample stated -- and the bookkeeping that ties the quoted counters to kernel sources."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(os.path.join(ROOT, "profiles", "r6_bench_default.json")) as f:
        return json.load(f)


def test_committed_bench_line_has_the_contract_fields():
    d = _line()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None                      # BASELINE.md holds no published number for this metric
    assert d["dtype"] == "f64" and "synthetic" in d["data"]
    assert "workload" in d["config"] and "model" not in d["config"]
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert d["metric"].split(" at ")[0] in base["metric"]          # "partition-state assignments/sec"
    assert "1048576" in d["config"]["workload"].replace(",", "") and "4096" in d["config"]["workload"].replace(",", "")
    # value is whole-job throughput of the timed region: 3 sweeps x 1,048,576 partitions x 3 state slots ... per call
    assert d["matches_oracle_digest"] is True
    per_call = d["value"] * d["ms_per_step"] * 1e-3
    assert abs(per_call - round(per_call)) < 1e-3 * per_call and per_call >= 1048576


def test_roofline_follows_from_its_inputs():
    r = _line()["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.0 < r["frac"] <= 1.0
    assert r["traffic"] is None or r["traffic"] > 0
    for k in _line()["roofline_per_kernel"]:
        assert abs(k["frac"] - k["achieved"] / k["peak"]) < 1e-9 and k["frac"] <= 1.0
        # achieved = algorithmic bytes per launch / average launch duration
        assert abs(k["achieved"] - k["algorithmic_bytes_per_launch"] / (k["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * k["achieved"]


def test_cpu_baseline_states_its_sample():
    c = _line()["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["value"] > 0 and c["sample"]
    assert c["unit"] == _line()["unit"]


def test_quoted_counters_are_tied_to_kernel_sources():
    """bench.py quotes a committed PMC profile only for the kernel sources it was taken from -- no list of "equivalent"
    sources: any other hash gives traffic null (the default run measures its traffic live anyway, bench.live_pmc)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench
    import profile_summary
    now = profile_summary.source_hash()
    for name in ("r6_pmc_hbm_config3.json", "r6_pmc_sq_config3.json", "r6_pmc_hbm_config5.json"):
        data, src = bench.profile_json(name)
        with open(os.path.join(ROOT, "profiles", name)) as f:
            profiled = json.load(f)["source_hash"]
        if now == profiled:
            assert data is not None and profiled in src
        else:
            assert data is None and "other kernel sources" in src
    data, src = bench.profile_json("r6_no_such_profile.json")
    assert data is None


def test_committed_line_measured_its_traffic_in_the_same_run():
    """The committed default line: roofline.traffic comes from PMC passes of the run itself, per launch of the dominant
    kernel, and both byte models stand side by side."""
    r = _line()["roofline"]
    assert r["traffic"] and "measured in this run" in r["traffic_from"]
    assert r["survey_8d_bytes_per_launch"] > r["algorithmic_bytes_per_launch"]
    assert abs(r["survey_8d_frac"] - r["survey_8d_GBps"] / r["peak"]) < 1e-9
    gr = _line()["general_regime"]
    assert len(gr) == 2 and all(w["matches_oracle_digest"] is True and w["headline"] is False for w in gr)


def test_committed_line_carries_every_baseline_config_and_the_boundary():
    """Round 5: the default line times BASELINE configs 2 and 5 as well (digests checked), names ONE kernel in `roofline`,
    runs the CPU oracle on all of config 3, and reports the transfers in steady state both ways."""
    d = _line()
    oc = d["other_configs"]
    assert [w["config"] for w in oc] == [2, 5, 5]
    assert all("error" not in w and w["matches_oracle_digest"] is True for w in oc), oc
    assert all(abs(w["roofline"]["frac"] - w["roofline"]["achieved"] / w["roofline"]["peak"]) < 1e-9 for w in oc)
    # (ONE kernel: the pass kernel with the largest share -- since round 6 the second sweep's chain pass and the third sweep's
    # k_stay_by_top pass are within a few per cent of each other)
    assert d["roofline"]["kernel"].startswith(("k_pass_chain<2,2,false>", "k_stay_by_top"))
    assert not any("/" in k["kernel"].split("(")[0] for k in d["roofline_per_kernel"] if k["kernel"].startswith("k_"))
    assert d["cpu_baseline"]["extrapolated"] is False and "NOT extrapolated" in d["cpu_baseline"]["sample"]
    t = d["transfers"]
    assert t["same_digest_both_ways"] is True
    assert t["page_locked"]["upload_s"] + t["page_locked"]["download_s"] < 4e-3          # <= 4 ms at config 3
    assert t["value_incl_transfers"] >= 300e6
    assert t["value_incl_transfers"] < d["value"]                                      # never the headline


def test_round6_line_items():
    """VERDICT r5 item 5 and what came with it, on the committed r6 line: a CPU figure beside EVERY GPU figure (config 5: the C
    oracle on a labelled sample), config 5 timed with a warm-up and two steps, no per-step issue figures for a pass that is not
    walked, `dense_scan_executed: false` wherever the 8(d) fraction exceeds 1, the one-rank RCCL leg with its per-collective
    latency, the replica mode on one GPU, the blocks' seconds -- and the whole default run inside a few minutes."""
    d = _line()
    oc = d["other_configs"]
    c5 = [w for w in oc if w["config"] == 5]
    assert len(c5) == 2 and all(w["warmup"] >= 1 and w["steps"] >= 2 for w in c5)
    cb = c5[1]["cpu_baseline"]
    assert cb["extrapolated"] is True and cb["cores"] == 1 and cb["kind"] == "port" and cb["value"] > 0 and "1/32" in cb["sample"]
    c2 = [w for w in oc if w["config"] == 2][0]
    assert c2["cpu_baseline"]["value"] > 0 and c2["cpu_baseline"]["extrapolated"] is False
    for k in d["roofline_per_kernel"] + [d["roofline"]]:
        if k.get("survey_8d_frac", 0) > 1.0:
            assert k["dense_scan_executed"] is False, k["kernel"]
        if "periodic" in k.get("kernel", "") or "all-blank" in k.get("kernel", ""):
            cp = k.get("critical_path") or {}
            assert "instructions_per_step_per_wave" not in cp and "cycles_per_instruction" not in cp
    assert d["roofline"]["survey_8d_whole_call"]["dense_scan_executed"] is False
    r = d["rccl_one_rank"]
    assert "error" not in r and r["rccl_world_size"] == 1 and r["same_digest_as_unsharded_plan"] is True
    assert r["comm_calls_per_plan"] == 2 * r["sweeps_per_call"] and 0.5 < r["us_per_collective"] < 1000
    rep = d["replicas_on_one_gpu"]
    assert [b["config"] for b in rep] == [3, 5] and all(b["headline"] is False for b in rep)
    for b in rep:
        runs = [x for x in b["runs"] if "aggregate_value" in x]
        assert [x["R"] for x in runs][:3] == [1, 4, 16] and all(x["every_digest_is_the_oracles"] is True for x in runs)
        assert all("replicas x%d on 1 GPU" % x["R"] == x["parallelism"] for x in runs)
    r5 = {x["R"]: x for x in rep[1]["runs"] if "aggregate_value" in x}
    assert r5[16]["aggregate_value"] > 8 * r5[1]["aggregate_value"]             # sixteen plans at (nearly) the latency of one
    assert max(x["aggregate_value"] for x in rep[1]["runs"] if "aggregate_value" in x) < d["value"]     # never the headline
    bs = d["block_seconds"]
    assert bs["total"] < 420 and {"headline", "cpu_baseline", "other_configs"} <= set(bs)
    assert d["ms_per_step"] <= 1.6                                                # VERDICT r5 item 4 (2.909 ms in round 5; target <= 2.4)
    assert 1 <= d["host_syncs_per_call"] <= 6                                     # (13 in round 5: DESIGN.md 4.5 "The host's shortcuts")


def test_live_line_of_a_rehearsal_has_the_same_fields():
    """bench.py itself, N = 1, on the CPU (BLANCE_BENCH_REHEARSAL: emulated kernels; the numbers mean nothing): the line it
    prints carries the contract's fields, so an edit of bench.py cannot drop one unnoticed until the next device run."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_simt_emulated import build_emu
    env = dict(os.environ, BLANCE_BENCH_REHEARSAL=build_emu())
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--parts", "600", "--nodes", "256", "--steps", "2",
                          "--warmup", "1", "--no-sharded"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    committed = _line()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["rehearsal"] is True and d["steps"] == 2 and d["warmup"] == 1 and d["n_gpus"] == 1
    assert d["metric"] == committed["metric"] and d["unit"] == committed["unit"] and d["dtype"] == committed["dtype"]
    # (dense_scan_executed stands only beside an 8(d) fraction above 1: the full-size line has it, a 600-partition rehearsal need not)
    assert set(committed["roofline"]) <= set(d["roofline"]) | {"traffic_from", "traffic_launches_counted", "critical_path", "occupancy",
                                                               "dense_scan_executed"}
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["sample"]


def test_bench_without_the_periodic_form_is_not_a_headline():
    """bench.py --no-periodic (every step of the all-blank chain pass walked) on the CPU rehearsal: same result as the default
    run of the same shape, the line says it is not the default; the default line carries both byte models of the roofline
    block and the general_regime block (two more workloads of the same size)."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_simt_emulated import build_emu
    env = dict(os.environ, BLANCE_BENCH_REHEARSAL=build_emu())
    got = {}
    for flag in ([], ["--no-periodic", "--no-extra"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--parts", "4096", "--nodes", "256", "--steps", "1",
                              "--warmup", "0", "--no-sharded", "--no-cpu-baseline"] + flag, env=env, capture_output=True,
                             text=True, timeout=900)
        assert out.returncode == 0, out.stdout + out.stderr
        got[bool(flag)] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert got[True]["result_sha256"] == got[False]["result_sha256"]
    assert "not_default" in got[True]["config"] and got[True]["config"]["headline"] is False
    assert "not_default" not in got[False]["config"]
    roof = got[False]["roofline"]
    for key in ("frac", "survey_8d_frac", "survey_8d_bytes_per_launch", "algorithmic_bytes_per_launch", "traffic", "traffic_from", "note"):
        assert key in roof, key
    assert roof["survey_8d_bytes_per_launch"] > roof["algorithmic_bytes_per_launch"]
    gr = got[False]["general_regime"]
    assert len(gr) == 2 and all("error" not in w and w["headline"] is False and w["ms_per_step"] > 0 for w in gr), gr
    assert "general_regime" not in got[True]
