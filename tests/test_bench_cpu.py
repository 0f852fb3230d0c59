"""CPU tests of bench.py's host logic and of the one-process-per-GPU placement helpers (no GPU, no library calls):
the compact `summary` the driver's 2 000-character tail must contain, the `--gpus` / launcher rule, NUMA pinning on a
fake sysfs topology."""
import importlib
import json
import os
import subprocess
import sys

from conftest import ROOT

S = importlib.import_module("openai_whisper_coreml_amd.sharding")


def _recorded_line():
    return json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_n1_driver_cmd.json")).read().strip().splitlines()[-1])


def test_summary_is_last_and_fits_the_drivers_tail():
    """the driver keeps the last 2 000 characters of the line; `summary` (<= 1 500 characters) is the LAST
    key, so the stage fractions, value_batch8, the token checks and one number per other configuration survive."""
    sys.path.insert(0, ROOT)
    import bench
    line = _recorded_line()
    line.pop("summary", None)
    line["summary"] = bench.build_summary(line)
    text = json.dumps(line)
    sm = json.dumps(line["summary"])
    assert len(sm) <= 1500, len(sm)
    assert list(line)[-1] == "summary" and text.endswith(sm + "}") and sm in text[-2000:]
    s = line["summary"]
    assert s["value"] == round(line["value"], 1) and s["value_batch8"] == round(line["value_batch8"], 1)
    assert set(s["stage_frac"]) == {"frontend", "encoder_xkv", "decode"}
    assert abs(s["stage_frac"]["decode"] - line["stage_roofline"]["decode"]["frac"]) < 1e-4
    assert abs(s["roofline"]["in_situ_frac"] - line["roofline"]["in_situ"]["frac"]) < 1e-4
    assert s["token_checks"] == line["token_checks"]
    assert set(s["other"]) == set(line["other_configs"])
    assert s["other"]["base_b32_one_group"][0] == round(line["other_configs"]["base_b32_one_group"]["value"], 1)
    # a line without the optional parts (N > 1: no other_configs; profiling flags: no roofline) still summarises
    bare = {k: v for k, v in line.items() if k not in ("other_configs", "roofline", "cpu_baseline", "early_stop", "summary")}
    b = bench.build_summary(bare)
    assert b["other"] == {} and b["roofline"]["frac"] is None and b["cpu_baseline"] is None


def test_gpus_flag_and_launcher_rule():
    """`--gpus` defaults to the launcher's WORLD_SIZE (torchrun --nproc-per-node 8 bench.py must work); an
    explicit value that disagrees is refused BEFORE any GPU work (so this runs on the CPU box)."""
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                       text=True, timeout=120, env=env, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr, r.stderr[-500:]
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"--gpus", type=int, default=None' in src and "--standalone" in src and "bind((" not in src


def _fake_sysfs(tmp_path, gpus, nodes):
    """gpus: {bdf: numa_node}; nodes: {node: cpulist}"""
    for bdf, n in gpus.items():
        d = tmp_path / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "numa_node").write_text("%d\n" % n)
    for n, cl in nodes.items():
        d = tmp_path / "devices" / "system" / "node" / ("node%d" % n)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cl + "\n")
    return str(tmp_path)


def test_numa_placement_on_a_fake_eight_gpu_node(tmp_path):
    bdfs = ["0000:%02x:00.0" % (5 + 16 * i) for i in range(8)]
    root = _fake_sysfs(tmp_path, {b: (0 if i < 4 else 1) for i, b in enumerate(bdfs)}, {0: "0-47,96-143", 1: "48-95,144-191"})
    assert S._parse_cpulist("0-3,8,10-11") == [0, 1, 2, 3, 8, 10, 11]
    shares = [S.numa_cpus_for_rank(bdfs, r, sysfs_root=root) for r in range(8)]
    node0 = set(S._parse_cpulist("0-47,96-143"))
    for r, sh in enumerate(shares):
        assert len(sh) == 24 and (set(sh) <= node0) == (r < 4), (r, sh[:4])
    for a in range(8):
        for b in range(a + 1, 8):
            assert not set(shares[a]) & set(shares[b])           # ranks never share a core
    # the process's own affinity (a cgroup / taskset) is respected
    assert S.numa_cpus_for_rank(bdfs, 0, allowed=range(0, 16), sysfs_root=root) == [0, 1, 2, 3]
    # too few cores to split: the ranks of the node share what there is; none at all -> do not pin
    assert S.numa_cpus_for_rank(bdfs, 1, allowed=range(0, 6), sysfs_root=root) == [0, 1, 2, 3, 4, 5]
    assert S.numa_cpus_for_rank(bdfs, 1, allowed=range(0, 2), sysfs_root=root) is None
    # numa_node = -1 (VMs, single-node boxes) or an unknown device: no pinning
    root2 = _fake_sysfs(tmp_path / "b", {"0000:05:00.0": -1}, {0: "0-7"})
    assert S.numa_cpus_for_rank(["0000:05:00.0"], 0, sysfs_root=root2) is None
    assert S.numa_cpus_for_rank(["0000:aa:00.0"], 0, sysfs_root=root2) is None
    assert S.numa_node_of_pci("0000:05:00.0", root) == 0
