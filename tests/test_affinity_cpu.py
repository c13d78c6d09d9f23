"""CPU placement of data-parallel ranks (parler_tts_amd/affinity.py): the planner on synthetic topologies, the
sysfs walk on a fake tree, and the 8-rank launcher rehearsal with the masks on the N > 1 line.  The reference's multi-process
pattern is training/run_parler_tts_training.py:1555, 1684-1689 (placement left to its launcher)."""
import json
import os
import subprocess
import sys

import pytest

from parler_tts_amd import affinity as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def fake_sysfs(root, gpu_cpulists, cpu_nodes=2):
    """KFD topology with `cpu_nodes` CPU nodes in front of the GPUs (as on a real host) + one PCI device per GPU"""
    nodes = os.path.join(root, "sys/class/kfd/kfd/topology/nodes")
    for i in range(cpu_nodes):
        os.makedirs(os.path.join(nodes, str(i)))
        open(os.path.join(nodes, str(i), "properties"), "w").write("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for g, (cpus, numa) in enumerate(gpu_cpulists):
        d = os.path.join(nodes, str(cpu_nodes + g))
        os.makedirs(d)
        bus = 0x05 + 0x10 * g
        open(os.path.join(d, "properties"), "w").write(f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {bus << 8}\ndomain 0\nname gfx950\n")
        pdir = os.path.join(root, "sys/bus/pci/devices", f"0000:{bus:02x}:00.0")
        os.makedirs(pdir)
        open(os.path.join(pdir, "local_cpulist"), "w").write(cpus + "\n")
        open(os.path.join(pdir, "numa_node"), "w").write(f"{numa}\n")


def test_cpulist_round_trip():
    assert A.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert A.format_cpulist([11, 0, 1, 2, 3, 8, 10]) == "0-3,8,10-11"
    assert A.parse_cpulist("") == [] and A.format_cpulist([]) == ""


def test_planner_gives_ranks_of_one_node_disjoint_slices_and_covers_the_node():
    node0, node1 = list(range(0, 64)) + list(range(128, 192)), list(range(64, 128)) + list(range(192, 256))
    gpu_cpus = [node0] * 4 + [node1] * 4                      # 8 GPUs, 4 per socket, SMT siblings in the upper half of the list
    allowed = list(range(256))
    plans = [A.plan_affinity(r, gpu_cpus, allowed) for r in range(8)]
    for r, p in enumerate(plans):
        assert len(p) == 32 and set(p) <= set(gpu_cpus[r])
    for a in range(8):
        for b in range(a + 1, 8):
            assert not set(plans[a]) & set(plans[b])
    assert sorted(sum(plans[:4], [])) == sorted(node0) and sorted(sum(plans[4:], [])) == sorted(node1)
    # with the SMT topology known a physical core's two threads stay in one slice (by position, ranks 0 and 2 would share cores)
    core_of = {c: c % 128 for c in range(256)}
    plans = [A.plan_affinity(r, gpu_cpus, allowed, core_of=core_of) for r in range(8)]
    assert plans[0] == list(range(0, 16)) + list(range(128, 144)) and plans[3] == list(range(48, 64)) + list(range(176, 192))
    assert plans[4] == list(range(64, 80)) + list(range(192, 208))
    for a in range(8):
        assert len(plans[a]) == 32 and {core_of[c] for c in plans[a]} == {core_of[c] for c in plans[a] if c < 128}
        for b in range(a + 1, 8):
            assert not {core_of[c] for c in plans[a]} & {core_of[c] for c in plans[b]}
    # a cpuset smaller than the node: slices of what is allowed
    plans = [A.plan_affinity(r, gpu_cpus, list(range(0, 16))) for r in range(4)]
    assert [len(p) for p in plans] == [4, 4, 4, 4] and sorted(sum(plans, [])) == list(range(16))
    # too few cores for slices of MIN_CORES: every rank of the node gets the whole node (equal per node)
    assert [A.plan_affinity(r, [[0, 1, 2, 3]] * 4, [0, 1, 2, 3]) for r in range(4)] == [[0, 1, 2, 3]] * 4
    # none of the GPU's cores allowed: nothing to bind to
    assert A.plan_affinity(0, [[8, 9]], [0, 1]) == []


def test_thread_siblings_come_from_the_cpu_topology(tmp_path):
    for c, sib in ((0, "0,4"), (4, "0,4"), (1, "1,5"), (5, "1,5")):
        d = tmp_path / f"sys/devices/system/cpu/cpu{c}/topology"
        d.mkdir(parents=True)
        (d / "thread_siblings_list").write_text(sib + "\n")
    assert A.thread_siblings([0, 1, 4, 5, 9], str(tmp_path)) == {0: 0, 4: 0, 1: 1, 5: 1, 9: 9}      # cpu9: unreadable -> itself
    assert A.plan_affinity(1, [[0, 1, 4, 5]] * 2, [0, 1, 4, 5], core_of=A.thread_siblings([0, 1, 4, 5], str(tmp_path))) == [1, 5]


def test_sysfs_walk_binds_this_process_and_opts_out(tmp_path, monkeypatch):
    before = sorted(os.sched_getaffinity(0))
    if len(before) < 4:
        pytest.skip("needs 4 cores")
    half = before[: len(before) // 2]
    fake_sysfs(str(tmp_path), [(A.format_cpulist(half), 0), (A.format_cpulist(before[len(before) // 2:]), 1)])
    assert A.gpu_pci_addresses(str(tmp_path)) == ["0000:05:00.0", "0000:15:00.0"]
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "PTTS_NO_AFFINITY"):
        monkeypatch.delenv(var, raising=False)
    try:
        monkeypatch.setenv("PTTS_NO_AFFINITY", "1")
        rec = A.bind_rank_to_numa_node(0, 2, sysfs=str(tmp_path))
        assert rec["bound"] is False and sorted(os.sched_getaffinity(0)) == before
        monkeypatch.delenv("PTTS_NO_AFFINITY")
        monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,0")          # device 0 of this process is the second GPU
        rec = A.bind_rank_to_numa_node(0, 2, sysfs=str(tmp_path))
        assert rec["bound"] is True and rec["gpu_pci"] == "0000:15:00.0" and rec["numa_node"] == 1
        assert sorted(os.sched_getaffinity(0)) == before[len(before) // 2:] and rec["n_cpus"] == len(before) - len(half)
    finally:
        os.sched_setaffinity(0, before)
    # no KFD topology (this container, any CPU box): unbound with a reason, never an exception
    rec = A.bind_rank_to_numa_node(0, 1, sysfs=str(tmp_path / "nothing"))
    assert rec["bound"] is False and "why" in rec and sorted(os.sched_getaffinity(0)) == before


def test_eight_rank_rehearsal_carries_masks_and_the_generate_spread(tmp_path):
    """bench.py --gpus 8 on CPU ranks over a fake 2-socket topology: every rank binds itself before its first GPU call would
    happen, the masks are equal per node or disjoint, and the line (still a valid rehearsal line) carries them next to the
    per-rank generate() times."""
    cores = sorted(os.sched_getaffinity(0))
    if len(cores) < 4:
        pytest.skip("needs 4 cores")
    n0, n1 = cores[: len(cores) // 2], cores[len(cores) // 2:]
    fake_sysfs(str(tmp_path), [(A.format_cpulist(n0), 0)] * 4 + [(A.format_cpulist(n1), 1)] * 4)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "PTTS_SHARE_GPU",
                                                           "PTTS_DIST_BACKEND", "PTTS_NO_AFFINITY", "HIP_VISIBLE_DEVICES",
                                                           "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")}
    env["PTTS_AFFINITY_SYSFS"] = str(tmp_path)
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--plumbing-only", "--batch", "2", "--max-length", "12", "--steps", "1",
                        "--warmup", "1"], capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    dp = line["data_parallel"]
    assert dp["ranks"] == 8 and dp["equals_single_gpu"] is True and "differ" not in line.get("invalid", "")
    masks = dp["affinity_by_rank"]
    assert len(masks) == 8 and all(m["bound"] for m in masks), masks
    sets = [set(A.parse_cpulist(m["cpus"])) for m in masks]
    for r_, s in enumerate(sets):
        assert s and s <= set(n0 if r_ < 4 else n1)
    for a in range(8):
        for b in range(a + 1, 8):
            same_node = (a < 4) == (b < 4)
            assert not (sets[a] & sets[b]) or (same_node and sets[a] == sets[b])      # disjoint, or equal per node
    assert [m["numa_node"] for m in masks] == [0] * 4 + [1] * 4
    assert len(dp["generate_ms_last_pass_by_rank"]) == 8 and "generate_ms_spread_pct" in dp
    # opting out leaves every rank where it was
    env["PTTS_NO_AFFINITY"] = "1"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plumbing-only", "--batch", "2", "--max-length", "12", "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert [m["bound"] for m in line["data_parallel"]["affinity_by_rank"]] == [False, False]


def test_pci_bus_walk_when_the_kfd_topology_is_unreadable(tmp_path, monkeypatch):
    """Some sandboxes deny the KFD topology (the pool's one-GPU boxes do, from Python): the walk falls back to AMD display /
    accelerator functions in PCI address order and the record says so."""
    before = sorted(os.sched_getaffinity(0))
    if len(before) < 2:
        pytest.skip("needs 2 cores")
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "PTTS_NO_AFFINITY"):
        monkeypatch.delenv(var, raising=False)
    base = tmp_path / "sys/bus/pci/devices"
    for bdf, vendor, cls, cpus in (("0000:05:00.0", "0x1002", "0x120000", A.format_cpulist(before[:1])),
                                   ("0000:03:00.0", "0x8086", "0x020000", A.format_cpulist(before)),          # a NIC: skipped
                                   ("0000:85:00.0", "0x1002", "0x038000", A.format_cpulist(before[1:]))):
        d = base / bdf
        d.mkdir(parents=True)
        (d / "vendor").write_text(vendor + "\n")
        (d / "class").write_text(cls + "\n")
        (d / "local_cpulist").write_text(cpus + "\n")
        (d / "numa_node").write_text("0\n")
    assert A.gpu_pci_addresses_by_bus(str(tmp_path)) == ["0000:05:00.0", "0000:85:00.0"]
    try:
        rec = A.bind_rank_to_numa_node(1, 2, sysfs=str(tmp_path))
        assert rec["walk"] == "pci-bus-order" and rec["gpu_pci"] == "0000:85:00.0" and rec["bound"] is True
        assert sorted(os.sched_getaffinity(0)) == before[1:]
    finally:
        os.sched_setaffinity(0, before)
