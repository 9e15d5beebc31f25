"""Where a rank runs on the host: the CPUs and the memory next to its GPU.

The decode path is fed over PCIe from page-locked host memory.  On a host with more than
one NUMA node, a rank whose threads and buffers sit on the far node pulls every
compressed byte across the socket link, so each rank puts itself on the CPUs of its
GPU's node *before* it allocates its page-locked buffers (the kernel places the pages
of a new allocation on the node of the thread that makes it).  The generator workers and
the library's parse threads are started afterwards and inherit the placement.

Two sources name the node of a GPU, both plain sysfs text:
  * /sys/bus/pci/devices/<bdf>/numa_node and local_cpulist, with <bdf> from the library
    (mj_device_pci_bus_id) once a context exists;
  * before anything has touched the GPU (worker processes are forked first): the KFD
    topology, /sys/class/kfd/kfd/topology/nodes/*/properties, whose GPU nodes are in the
    order the HIP runtime numbers them; ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES given
    as plain indices are applied on top.
A host that tells nothing (numa_node -1: one node, or a virtual machine) gets an even
split of the CPUs the process may use, so that the ranks' workers do not pile up on the
same cores.

The reference has no counterpart: it is one thread on one CPU (jpeg_decoder.c:29-112).
"""
from __future__ import annotations

import glob
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Set

SYSFS = os.environ.get("MJ_SYSFS_ROOT", "/sys")      # another root: a made-up host in the tests


def parse_cpulist(text: str) -> List[int]:
    """ "0-3,8,10-11" -> [0, 1, 2, 3, 8, 10, 11] (the kernel's list format)."""
    out: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-", 1)
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return sorted(set(out))


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def pci_numa(bdf: str, sysfs: str = SYSFS):
    """(numa node or -1, local CPUs or []) of the PCI device `bdf` ("0000:0d:00.0")."""
    base = os.path.join(sysfs, "bus", "pci", "devices", bdf.lower())
    node = _read(os.path.join(base, "numa_node"))
    cpus = _read(os.path.join(base, "local_cpulist"))
    try:
        n = int(node) if node is not None else -1
    except ValueError:
        n = -1
    try:
        c = parse_cpulist(cpus) if cpus else []
    except ValueError:
        c = []
    return n, c


def kfd_gpu_bdfs(sysfs: str = SYSFS) -> List[str]:
    """PCI addresses of the GPUs in KFD topology order (the order the runtime numbers
    them before the VISIBLE_DEVICES variables are applied).  [] when there is no KFD."""
    nodes = []
    for d in glob.glob(os.path.join(sysfs, "class", "kfd", "kfd", "topology", "nodes", "*")):
        try:
            idx = int(os.path.basename(d))
        except ValueError:
            continue
        text = _read(os.path.join(d, "properties"))
        if not text:
            continue
        prop: Dict[str, int] = {}
        for line in text.splitlines():
            kv = line.split()
            if len(kv) == 2:
                try:
                    prop[kv[0]] = int(kv[1])
                except ValueError:
                    pass
        if prop.get("simd_count", 0) <= 0:
            continue                      # a CPU node
        loc = prop.get("location_id", 0)
        nodes.append((idx, "%04x:%02x:%02x.%x" % (prop.get("domain", 0), (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7)))
    return [b for _, b in sorted(nodes)]


def _visible(bdfs: Sequence[str], env: Dict[str, str]) -> List[str]:
    """Apply ROCR_VISIBLE_DEVICES, then HIP_VISIBLE_DEVICES, where they are lists of plain
    indices (UUID forms are left alone: the guess is then checked against the library)."""
    out = list(bdfs)
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        v = env.get(var, "").strip()
        if not v:
            continue
        try:
            idx = [int(x) for x in v.split(",") if x.strip() != ""]
        except ValueError:
            return out
        out = [out[i] for i in idx if 0 <= i < len(out)]
    return out


@dataclass
class Placement:
    local_rank: int
    local_world: int
    bdf: Optional[str]            # PCI address of the rank's GPU, None if unknown
    numa_node: int                # -1: the host does not say
    cpus: List[int]               # what the rank is (to be) bound to
    source: str                   # "pci" (library), "kfd" (sysfs guess) or "split" (no NUMA information)
    applied: bool = False
    note: str = ""
    shared_with: List[int] = field(default_factory=list)   # local ranks on the same node, this one included

    def as_dict(self):
        return {"local_rank": self.local_rank, "bdf": self.bdf, "numa_node": self.numa_node, "n_cpus": len(self.cpus),
                "cpus": _fmt_cpulist(self.cpus), "source": self.source, "applied": self.applied, "note": self.note}


def _fmt_cpulist(cpus: Sequence[int]) -> str:
    cpus = sorted(cpus)
    parts, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        parts.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(parts)


def _share(cpus: List[int], k: int, m: int, min_cpus: int = 2) -> List[int]:
    """The k-th of m contiguous shares of `cpus`; the whole list when a share would be
    smaller than min_cpus."""
    if m <= 1 or len(cpus) // m < min_cpus:
        return list(cpus)
    lo, hi = len(cpus) * k // m, len(cpus) * (k + 1) // m
    return cpus[lo:hi]


def plan(local_rank: int, local_world: int, allowed: Optional[Set[int]] = None, bdfs: Optional[Sequence[str]] = None,
         source: str = "kfd", sysfs: str = SYSFS, env: Optional[Dict[str, str]] = None) -> Placement:
    """The CPUs for local rank `local_rank` of `local_world` ranks, rank r on GPU r.

    bdfs: PCI address of every local rank's GPU (index = local rank) when known; else the
    KFD topology is read.  allowed: the CPUs the process may use (default: its affinity
    mask).  Ranks whose GPUs share a node divide that node's allowed CPUs evenly, in
    order of local rank; a rank whose node is unknown, or has none of the allowed CPUs,
    gets its share of all allowed CPUs."""
    if allowed is None:
        allowed = set(os.sched_getaffinity(0))
    allowed_l = sorted(allowed)
    if bdfs is None:
        bdfs = _visible(kfd_gpu_bdfs(sysfs), dict(os.environ) if env is None else env)
    nodes, lists = [], []
    for r in range(local_world):
        if r < len(bdfs) and bdfs[r]:
            n, c = pci_numa(bdfs[r], sysfs)
        else:
            n, c = -1, []
        c = [x for x in c if x in allowed]
        if n < 0 or not c:
            # no node, or none of its CPUs are ours: fall back to the even split below
            n, c = -1, []
        nodes.append(n)
        lists.append(c)
    me = local_rank
    bdf = bdfs[me] if me < len(bdfs) else None
    if nodes[me] >= 0:
        peers = [r for r in range(local_world) if nodes[r] == nodes[me]]
        cpus = _share(lists[me], peers.index(me), len(peers))
        return Placement(me, local_world, bdf, nodes[me], cpus, source, shared_with=peers)
    peers = [r for r in range(local_world) if nodes[r] < 0]
    # CPUs that no located rank's node claims, if any are left; else all allowed CPUs
    claimed = set(x for r in range(local_world) if nodes[r] >= 0 for x in lists[r])
    free = [x for x in allowed_l if x not in claimed] or allowed_l
    cpus = _share(free, peers.index(me), len(peers))
    return Placement(me, local_world, bdf, -1, cpus, "split",
                     note="the host names no NUMA node for this GPU: an even share of the allowed CPUs", shared_with=peers)


def apply(p: Placement) -> Placement:
    """Bind the calling process (and whatever it starts from now on) to p.cpus."""
    if not p.cpus:
        p.note = (p.note + "; " if p.note else "") + "no CPUs to bind to: affinity left as it was"
        return p
    try:
        os.sched_setaffinity(0, set(p.cpus))
        p.applied = True
    except OSError as e:
        p.note = (p.note + "; " if p.note else "") + f"sched_setaffinity refused: {e}"
    return p


def gpu_sysfs_dir(bdf: str, sysfs: str = SYSFS) -> Optional[str]:
    d = os.path.join(sysfs, "bus", "pci", "devices", bdf.lower())
    return d if os.path.isdir(d) else None


class GpuSensors:
    """Shader clock and board power of one GPU, read from sysfs text files
    (pp_dpm_sclk: the starred level; hwmon power1_input or power1_average: microwatts).
    Either may be missing; read() then leaves it out."""

    def __init__(self, bdf: str, sysfs: str = SYSFS):
        self.sclk_path = None
        self.power_path = None
        d = gpu_sysfs_dir(bdf, sysfs) if bdf else None
        if not d:
            return
        p = os.path.join(d, "pp_dpm_sclk")
        if os.access(p, os.R_OK):
            self.sclk_path = p
        for name in ("power1_input", "power1_average"):
            hits = glob.glob(os.path.join(d, "hwmon", "hwmon*", name))
            if hits and os.access(hits[0], os.R_OK):
                self.power_path = hits[0]
                break

    def read(self):
        out = {}
        if self.sclk_path:
            text = _read(self.sclk_path) or ""
            for line in text.splitlines():
                if line.rstrip().endswith("*"):
                    try:
                        out["sclk_MHz"] = int("".join(ch for ch in line.split(":", 1)[1] if ch.isdigit()))
                    except (ValueError, IndexError):
                        pass
        if self.power_path:
            try:
                out["power_W"] = int(_read(self.power_path)) / 1e6
            except (TypeError, ValueError):
                pass
        return out
