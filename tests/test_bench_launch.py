"""bench.py's launcher logic on CPU: `--gpus N` without a launcher spawns N ranks under torch.distributed.run (as a child process, before
anything touches a GPU), and a rank count that disagrees with --gpus is refused."""
import json
import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_workloads_name_every_gpu_config_of_baseline_json():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    named = sorted(w["config"] for w in bench.WORKLOADS.values())
    assert named == [1, 2, 4]                      # configs[0] is the CPU plumbing case (cpu_baseline.uniform_path_config0), configs[3] the training run
    assert len(base["configs"]) == 5
    a = bench.parse(["--workload", "garden"])
    assert (a.wl["H"], a.wl["W"], a.wl["model"]) == (840, 1297, "palette") and abs(a.wl["dt_gamma"] - 1 / 128) < 1e-12
    a = bench.parse([])
    assert (a.wl["H"], a.wl["W"], a.wl["model"], a.wl["dt_gamma"], a.gpus) == (800, 800, "nerf", 0.0, 1)
    assert bench.parse(["--model", "palette"]).workload == "lego_palette"


def test_more_than_one_rank_defaults_to_the_north_star_question(monkeypatch):
    """N > 1 without flags = configs[4]: ONE garden frame split over the ranks (strong scaling); N = 1 = configs[1]; explicit flags win."""
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    a = bench.parse(["--gpus", "8"])
    assert (a.workload, a.scaling, a.defaulted_for_ranks, a.wl["config"], a.wl["model"]) == ("garden", "strong", True, 4, "palette")
    a = bench.parse([])
    assert (a.workload, a.scaling, a.defaulted_for_ranks) == ("lego", "weak", False)
    for flags, want in ((["--workload", "lego"], ("lego", "weak")), (["--scaling", "weak"], ("lego", "weak")), (["--model", "palette"], ("lego_palette", "weak")),
                        (["--workload", "garden"], ("garden", "strong")), (["--workload", "garden", "--scaling", "weak"], ("garden", "weak"))):
        a = bench.parse(["--gpus", "4"] + flags)
        assert (a.workload, a.scaling) == want and not a.defaulted_for_ranks, flags
    a = bench.parse(["--dist-default"])                  # the same defaults over a one-rank communicator (the -m gpu test of the N > 1 line)
    assert (a.workload, a.scaling, a.gpus) == ("garden", "strong", 1)
    monkeypatch.setenv("WORLD_SIZE", "2")                # under the driver's launcher the rank count comes from the environment
    a = bench.parse(["--gpus", "2", "--steps", "5"])
    assert (a.workload, a.scaling) == ("garden", "strong")
    again = bench.parse(bench.core_argv(a))              # a child pass repeats the resolved workload and scaling, not the defaults of its own rank count
    monkeypatch.delenv("WORLD_SIZE")
    again1 = bench.parse(bench.core_argv(a))
    assert (again.workload, again.scaling) == (again1.workload, again1.scaling) == ("garden", "strong")


def test_gpus_flag_spawns_that_many_ranks(tmp_path, monkeypatch):
    """The spawn path really produces WORLD_SIZE == --gpus ranks (a stand-in script records what each rank sees)."""
    probe = tmp_path / "probe.py"
    probe.write_text("import os, sys\n"
                     "open(os.path.join(sys.argv[-1], 'rank' + os.environ['RANK']), 'w').write(os.environ['WORLD_SIZE'] + ' ' + os.environ['LOCAL_RANK'] + ' ' + ' '.join(sys.argv[1:-1]))\n")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    args = bench.parse(["--gpus", "2", "--steps", "3"])
    rc = bench.spawn_ranks(args, ["--gpus", "2", "--steps", "3", str(tmp_path)], script=str(probe))
    assert rc == 0
    seen = sorted(f for f in os.listdir(tmp_path) if f.startswith("rank"))
    assert seen == ["rank0", "rank1"]
    for r, f in enumerate(seen):
        world, local, *argv = (tmp_path / f).read_text().split()
        assert (world, local) == ("2", str(r)) and argv == ["--gpus", "2", "--steps", "3"]   # the ranks get the same flags


def test_world_from_env(monkeypatch):
    import torch
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.world_from_env(bench.parse([]), []) == (1, 0, 0)
    # --gpus 2, no launcher, 2 devices visible: the parent spawns and exits with the children's code; it never initialises a GPU itself
    calls = []
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.setattr(bench, "spawn_ranks", lambda a, argv, script=None: calls.append((a.gpus, list(argv))) or 7)
    with pytest.raises(SystemExit) as e:
        bench.world_from_env(bench.parse(["--gpus", "2"]), ["--gpus", "2"])
    assert e.value.code == 7 and calls == [(2, ["--gpus", "2"])]
    # fewer devices than ranks: refused loudly
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit, match="only 1 GPU"):
        bench.world_from_env(bench.parse(["--gpus", "2"]), ["--gpus", "2"])
    # under a launcher: ranks from the environment, and they must agree with --gpus
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "3")
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert bench.world_from_env(bench.parse(["--gpus", "4"]), []) == (4, 3, 3)
    with pytest.raises(SystemExit, match="must agree"):
        bench.world_from_env(bench.parse(["--gpus", "8"]), [])
    with pytest.raises(SystemExit, match="must agree"):
        bench.world_from_env(bench.parse([]), [])


def test_bench_helpers_without_a_gpu():
    """core_argv (what bench.py's rocprofv3 --pmc child passes repeat) round-trips through parse(); l2_bound_of is plain arithmetic."""
    import bench
    args = bench.parse(["--workload", "garden", "--density-scale", "0.5", "--field-precision", "fp32", "--static-pose", "--num-basis", "6", "--dt-gamma", "0.01"])
    again = bench.parse(bench.core_argv(args) + ["--steps", "3"])
    for k in ("workload", "density_scale", "field_precision", "static_pose", "num_basis", "ray_order", "fp16", "half_tables", "pred_clip", "no_interleave", "mode", "scene"):
        assert getattr(args, k) == getattr(again, k), k
    assert again.wl["dt_gamma"] == 0.01 and again.steps == 3
    b = bench.l2_bound_of(365482, 1, 0.0766)
    assert b["lane_requests_per_launch"] == 365482 * 128 and 0.9 < b["lane_requests_per_clk_per_cu"] < 1.1


def test_child_passes_of_a_rank_never_see_the_launcher(monkeypatch):
    """Round 6: rank 0 of an N > 1 run measures roofline.traffic in single-GPU child passes of this script.  Their environment carries none of the launcher's
    rank variables (parse() switches its defaults on WORLD_SIZE; a child must not try to join the parent's communicator), --emulate-shard makes them render
    rank 0's shard, and roofline_block prices a lookup whose algorithmic rate exceeds the HBM peak against the L2 bandwidth."""
    for k, v in (("WORLD_SIZE", "8"), ("RANK", "0"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29500"), ("PNR_BENCH_FORCE_DIST", "1"), ("TORCHELASTIC_RUN_ID", "x")):
        monkeypatch.setenv(k, v)
    env = bench.child_env(TMPDIR="/tmp")
    assert env["TMPDIR"] == "/tmp" and not any(k in env for k in bench.LAUNCHER_ENV)
    assert "PATH" in env
    a = bench.parse(["--gpus", "8"])
    monkeypatch.delenv("WORLD_SIZE")
    child = bench.parse(bench.core_argv(a) + ["--emulate-shard", "0/8", "--steps", "3"])
    assert (child.workload, child.scaling, child.shard, child.gpus) == ("garden", "strong", (0, 8), 1)
    with pytest.raises(SystemExit):
        bench.parse(["--emulate-shard", "8/8"])
    assert bench.parse(bench.core_argv(bench.parse(["--one-call-per-frame"]))).one_call_per_frame is True
    # the roofline object's bound
    hbm = bench.roofline_block("k", 6368.9, 3.04e8, {"x": 1}, 27, 27 * 0.0668, 27 * 365482, 1164, 1)
    assert hbm["bound"] == "hbm" and abs(hbm["frac"] - 6368.9 / 8000.0) < 1e-9 and 0.5 < hbm["hbm_frac_of_measured_traffic"] < 0.6 and 0.7 < hbm["traffic_over_algorithmic"] < 0.73
    l2 = bench.roofline_block("k", 17492.6, 6.16e8, None, 29, 29 * 0.1446, 29 * 2 * 1086526, 1164, 2)
    assert l2["bound"] == "l2" and l2["peak"] == bench.L2_PEAK_GBS and 0.5 < l2["frac"] < 0.52 and l2["algorithmic_over_hbm_peak"] > 2.0 and 0.5 < l2["hbm_frac_of_measured_traffic"] < 0.56
    assert l2["algorithmic_bytes_per_sample"] == 2328 and "note" in l2 and l2["closer_ceiling"]["name"].startswith("hbm") and l2["closer_ceiling"]["frac"] == l2["hbm_frac_of_measured_traffic"]
    none = bench.roofline_block("k", 100.0, None, None, 1, 1.0, 10, 1164, 1)
    assert none["traffic"] is None and "hbm_frac_of_measured_traffic" not in none
