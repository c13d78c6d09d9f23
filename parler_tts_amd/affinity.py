"""CPU placement of a data-parallel rank: bind the process to the cores next to ITS GPU before anything touches the card.

Why: the token loop has no collective, so a multi-GPU pass is as fast as its slowest rank, and a rank's
speed is set by its host loop - one thread that queues graph replays and watches a pinned-memory stamp every 16 steps.  Left to
the scheduler, eight such loops migrate between sockets and share cores with each other's runtime threads; DESIGN section 6
names that jitter as the only thing that decides the weak-scaling efficiency.  The reference leaves placement to `accelerate`'s
launcher (training/run_parler_tts_training.py:1555, 1684-1689 is its multi-process pattern); here the rank does it itself:

    GPU index -> KFD topology node -> PCI address -> /sys/bus/pci/devices/<bdf>/local_cpulist -> os.sched_setaffinity

Pure stdlib on purpose: no torch, no HIP call - it runs before the process has initialised a GPU and never execs a wrapper
(`taskset` / `numactl` in front of a GPU process is refused on the pool this is measured on).  `PTTS_NO_AFFINITY=1` opts out.
Ranks whose GPUs share a NUMA node get DISJOINT slices of that node's physical cores (both hardware threads of a core in the same
slice; at least MIN_CORES each), else the whole node; every failure leaves the affinity alone and says why in the returned record.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

MIN_CORES = 2          # a slice smaller than this (few cores, many ranks) falls back to the whole node's cores


def parse_cpulist(text: str) -> List[int]:
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            out.extend(range(int(lo), int(hi) + 1))
        else:
            out.append(int(part))
    return sorted(set(out))


def format_cpulist(cpus: Sequence[int]) -> str:
    cpus = sorted(set(cpus))
    runs, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        runs.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(runs)


def gpu_pci_addresses(sysfs: str = "/") -> List[str]:
    """PCI addresses of the GPUs in HIP's enumeration order = KFD topology nodes with SIMDs, by node id."""
    base = os.path.join(sysfs, "sys/class/kfd/kfd/topology/nodes")
    out = []
    for node in sorted((n for n in os.listdir(base) if n.isdigit()), key=int):
        props: Dict[str, int] = {}
        with open(os.path.join(base, node, "properties")) as f:
            for line in f:
                k, _, v = line.strip().partition(" ")
                if v.strip().lstrip("-").isdigit():
                    props[k] = int(v)
        if props.get("simd_count", 0) <= 0:
            continue                                    # a CPU node
        loc, dom = props.get("location_id", 0), props.get("domain", 0)
        out.append(f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7}")
    return out


def gpu_pci_addresses_by_bus(sysfs: str = "/") -> List[str]:
    """Fallback when the KFD topology cannot be read (some sandboxes deny it): AMD display / accelerator functions in PCI address
    order - the order HIP enumerates them in on the nodes this was written for, but not a guarantee, which is why the record says
    which walk produced the binding."""
    base = os.path.join(sysfs, "sys/bus/pci/devices")
    out = []
    for bdf in sorted(os.listdir(base)):
        try:
            with open(os.path.join(base, bdf, "vendor")) as f:
                if f.read().strip().lower() != "0x1002":
                    continue
            with open(os.path.join(base, bdf, "class")) as f:
                cls = f.read().strip().lower()
        except OSError:
            continue
        if cls.startswith(("0x0300", "0x0302", "0x0380", "0x1200")):        # VGA / 3D / display / processing accelerator
            out.append(bdf)
    return out


def visible_gpu_indices(n_gpus: int, env: Optional[Dict[str, str]] = None) -> List[int]:
    """physical index of every visible device, honouring ROCR_VISIBLE_DEVICES then HIP_ / CUDA_VISIBLE_DEVICES (index lists)"""
    env = os.environ if env is None else env
    idx = list(range(n_gpus))
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(var)
        if v is None or not v.strip():
            continue
        try:
            sel = [int(x) for x in v.split(",") if x.strip() != ""]
        except ValueError:
            raise RuntimeError(f"{var}={v!r}: only index lists are understood here")
        idx = [idx[i] for i in sel if 0 <= i < len(idx)]
        if var != "ROCR_VISIBLE_DEVICES":
            break                                       # HIP_ and CUDA_ are the same filter under two names
    return idx


def thread_siblings(cpus: Sequence[int], sysfs: str = "/") -> Dict[int, int]:
    """cpu -> lowest-numbered hardware thread of its physical core (itself where the topology cannot be read)"""
    out: Dict[int, int] = {}
    for c in cpus:
        try:
            with open(os.path.join(sysfs, f"sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list")) as f:
                out[c] = min(parse_cpulist(f.read()) or [c])
        except (OSError, ValueError):
            out[c] = c
    return out


def plan_affinity(slot: int, gpu_cpus: Sequence[Sequence[int]], allowed: Sequence[int], min_cores: int = MIN_CORES,
                  core_of: Optional[Dict[int, int]] = None) -> List[int]:
    """Cores for the rank in `slot`.  gpu_cpus[i] = cores local to the GPU of slot i (one entry per rank of this node, in rank
    order), `allowed` = what the process may use at all (cgroup cpuset).  Ranks whose GPUs have the same local cores share
    them out in contiguous, disjoint slices by slot order; a slice under `min_cores` cores -> the whole set (equal per node).
    `core_of` (thread_siblings) keeps the hardware threads of a physical core in ONE slice: a node lists its SMT siblings in the
    upper half of its cpulist ("0-63,128-191"), and slicing that list by position would seat two ranks' host loops on the two
    threads of the same cores."""
    allowed_s = set(allowed)
    mine = [c for c in gpu_cpus[slot] if c in allowed_s]
    if not mine:
        return []
    key = tuple(sorted(gpu_cpus[slot]))
    peers = [i for i, c in enumerate(gpu_cpus) if tuple(sorted(c)) == key]
    k, pos = len(peers), peers.index(slot)
    core_of = core_of or {}
    groups: Dict[int, List[int]] = {}
    for c in sorted(mine):
        groups.setdefault(core_of.get(c, c), []).append(c)
    cores = [groups[g] for g in sorted(groups)]              # physical cores, each with its allowed hardware threads
    per = len(cores) // k
    if k == 1 or per == 0 or sum(len(g) for g in cores[:per]) < min_cores:
        return sorted(mine)
    lo = pos * per
    hi = len(cores) if pos == k - 1 else lo + per
    return sorted(c for g in cores[lo:hi] for c in g)


def bind_rank_to_numa_node(local_rank: int, local_world: int = 1, gpu_of_slot: Optional[Sequence[int]] = None,
                           sysfs: Optional[str] = None) -> Dict[str, object]:
    """Bind THIS process to the cores of its GPU's NUMA node (see the module text).  `gpu_of_slot[i]` = visible-device index rank
    slot i uses (default: slot i uses device i; a shared-card rehearsal passes [0] * local_world).  Returns the record that goes
    into the bench line: {"bound": bool, "cpus": "0-15", "n_cpus": 16, "numa_node": 0, "gpu_pci": "...", "why": "..."}."""
    rec: Dict[str, object] = {"bound": False, "cpus": format_cpulist(sorted(os.sched_getaffinity(0))),
                              "n_cpus": len(os.sched_getaffinity(0)), "numa_node": None, "gpu_pci": None}
    if os.environ.get("PTTS_NO_AFFINITY"):
        rec["why"] = "PTTS_NO_AFFINITY set"
        return rec
    sysfs = sysfs or os.environ.get("PTTS_AFFINITY_SYSFS") or "/"       # the variable: a fake tree for the CPU rehearsal test
    stage = "KFD topology"
    try:
        try:
            pci = gpu_pci_addresses(sysfs)
            rec["walk"] = "kfd"
        except OSError as kfd_exc:
            stage = f"PCI bus walk (KFD topology unreadable: {type(kfd_exc).__name__})"
            pci = gpu_pci_addresses_by_bus(sysfs)
            rec["walk"] = "pci-bus-order"
        vis = visible_gpu_indices(len(pci))
        slots = list(gpu_of_slot) if gpu_of_slot is not None else list(range(local_world))
        if local_rank >= len(slots) or any(s >= len(vis) for s in slots):
            rec["why"] = f"{len(vis)} visible GPU(s) in the KFD topology, rank slots {slots}"
            return rec
        bdf = pci[vis[slots[local_rank]]]
        rec["gpu_pci"] = bdf
        stage = "PCI local_cpulist"
        cpus = []
        for s in slots:
            with open(os.path.join(sysfs, "sys/bus/pci/devices", pci[vis[s]], "local_cpulist")) as f:
                cpus.append(parse_cpulist(f.read()))
        try:
            with open(os.path.join(sysfs, "sys/bus/pci/devices", bdf, "numa_node")) as f:
                rec["numa_node"] = int(f.read().strip())
        except (OSError, ValueError):
            pass
        allowed = sorted(os.sched_getaffinity(0))
        want = plan_affinity(local_rank, cpus, allowed, core_of=thread_siblings(cpus[local_rank], sysfs))
        rec["planned_cpus"] = format_cpulist(want)
        if not want:
            rec["why"] = f"none of the GPU's local cores ({format_cpulist(cpus[local_rank])}) is in this process's cpuset"
            return rec
        stage = "sched_setaffinity"        # a sandbox may refuse it (EPERM on the pool's one-GPU boxes): the rank then runs unbound
        os.sched_setaffinity(0, want)
        got = sorted(os.sched_getaffinity(0))
        rec.update(bound=True, cpus=format_cpulist(got), n_cpus=len(got))
    except Exception as exc:  # noqa: BLE001 - placement is an optimisation: never the reason a rank fails to start
        rec["why"] = f"{stage}: {type(exc).__name__}: {exc}"
    return rec
