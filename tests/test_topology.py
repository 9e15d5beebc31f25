"""Placement of ranks on the host (CPU checks against made-up sysfs trees)."""
import os

import pytest

from jpeg_decoder_amd import topology as T


def make_sysfs(root, gpus, nodes):
    """gpus: list of (bus, numa node or -1); nodes: {node: cpulist text}.  KFD node 0 is the CPU."""
    def put(path, text):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            f.write(text)
    kfd = os.path.join(root, "class", "kfd", "kfd", "topology", "nodes")
    put(os.path.join(kfd, "0", "properties"), "cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    bdfs = []
    for i, (bus, node) in enumerate(gpus):
        put(os.path.join(kfd, str(i + 1), "properties"),
            f"cpu_cores_count 0\nsimd_count 1024\nlocation_id {bus << 8}\ndomain 0\ndrm_render_minor {128 + i}\n")
        bdf = "0000:%02x:00.0" % bus
        bdfs.append(bdf)
        d = os.path.join(root, "bus", "pci", "devices", bdf)
        put(os.path.join(d, "numa_node"), f"{node}\n")
        put(os.path.join(d, "local_cpulist"), (nodes.get(node, "") if node >= 0 else ",".join(nodes.values())) + "\n")
        put(os.path.join(d, "pp_dpm_sclk"), "0: 500Mhz\n1: 2100Mhz *\n2: 2400Mhz\n")
        put(os.path.join(d, "hwmon", "hwmon3", "power1_input"), "750000000\n")
    return bdfs


def test_parse_cpulist():
    assert T.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert T.parse_cpulist("") == []
    assert T.parse_cpulist("5") == [5]
    assert T._fmt_cpulist([0, 1, 2, 3, 8, 10, 11]) == "0-3,8,10-11"


def test_kfd_order_and_visible_devices(tmp_path):
    root = str(tmp_path)
    bdfs = make_sysfs(root, [(0x0d, 0), (0x1d, 0), (0x8d, 1), (0x9d, 1)], {0: "0-31", 1: "32-63"})
    assert T.kfd_gpu_bdfs(root) == bdfs
    assert T._visible(bdfs, {"HIP_VISIBLE_DEVICES": "2,3"}) == bdfs[2:]
    assert T._visible(bdfs, {"ROCR_VISIBLE_DEVICES": "1,2,3", "HIP_VISIBLE_DEVICES": "1"}) == [bdfs[2]]
    assert T._visible(bdfs, {"HIP_VISIBLE_DEVICES": "GPU-abcdef"}) == bdfs       # not plain indices: left alone
    assert T.pci_numa(bdfs[2], root) == (1, list(range(32, 64)))
    assert T.pci_numa("0000:ff:00.0", root) == (-1, [])


def test_eight_gpus_on_two_nodes_get_disjoint_shares_of_their_own_node(tmp_path):
    root = str(tmp_path)
    gpus = [(0x10 + i, 0 if i < 4 else 1) for i in range(8)]
    make_sysfs(root, gpus, {0: "0-63", 1: "64-127"})
    allowed = set(range(128))
    seen = set()
    for r in range(8):
        p = T.plan(r, 8, allowed, sysfs=root, env={})
        assert p.source == "kfd" and p.numa_node == (0 if r < 4 else 1)
        assert len(p.cpus) == 16 and not (set(p.cpus) & seen)
        assert set(p.cpus) <= (set(range(64)) if r < 4 else set(range(64, 128)))
        assert p.shared_with == (list(range(4)) if r < 4 else list(range(4, 8)))
        seen |= set(p.cpus)
    assert seen == allowed
    # from the library's PCI addresses instead of the KFD guess, in another order
    bdfs = ["0000:%02x:00.0" % (0x17 - i) for i in range(8)]
    p = T.plan(0, 8, allowed, bdfs=bdfs, source="pci", sysfs=root)
    assert p.numa_node == 1 and p.bdf == "0000:17:00.0" and p.cpus == list(range(64, 80))


def test_cgroup_that_leaves_only_part_of_a_node(tmp_path):
    root = str(tmp_path)
    make_sysfs(root, [(0x0d, 0), (0x8d, 1)], {0: "0-31", 1: "32-63"})
    allowed = set(range(24, 40))            # 8 CPUs of each node
    assert T.plan(0, 2, allowed, sysfs=root, env={}).cpus == list(range(24, 32))
    assert T.plan(1, 2, allowed, sysfs=root, env={}).cpus == list(range(32, 40))
    # none of the GPU's node: an even share of what there is
    p = T.plan(1, 2, set(range(0, 8)), sysfs=root, env={})
    assert p.numa_node == -1 and p.source == "split" and set(p.cpus) <= set(range(8)) and p.cpus


def test_host_without_numa_information_splits_evenly(tmp_path):
    root = str(tmp_path)
    make_sysfs(root, [(0x0d, -1), (0x0e, -1), (0x0f, -1), (0x10, -1)], {0: "0-15"})
    got = [T.plan(r, 4, set(range(16)), sysfs=root, env={}) for r in range(4)]
    assert [p.cpus for p in got] == [list(range(4 * r, 4 * r + 4)) for r in range(4)]
    assert all(p.numa_node == -1 and p.source == "split" for p in got)
    # shares smaller than two CPUs are not cut: every rank keeps the whole list
    assert T.plan(3, 8, set(range(8)), sysfs=root, env={}).cpus == list(range(8))
    # no KFD at all (this container)
    assert T.plan(0, 1, {0, 1, 2}, sysfs=str(tmp_path / "none"), env={}).cpus == [0, 1, 2]


def test_apply_binds_the_process_and_children_inherit(tmp_path):
    before = os.sched_getaffinity(0)
    if len(before) < 2:
        pytest.skip("one CPU")
    try:
        keep = sorted(before)[:1]
        p = T.Placement(0, 1, None, -1, keep, "split")
        assert T.apply(p).applied and os.sched_getaffinity(0) == set(keep)
        pid = os.fork()
        if pid == 0:
            os._exit(0 if os.sched_getaffinity(0) == set(keep) else 1)
        assert os.waitpid(pid, 0)[1] == 0
        q = T.apply(T.Placement(0, 1, None, -1, [], "split"))
        assert not q.applied and "left as it was" in q.note
    finally:
        os.sched_setaffinity(0, before)


def test_sensors_read_clock_and_power(tmp_path):
    root = str(tmp_path)
    bdfs = make_sysfs(root, [(0x0d, 0)], {0: "0-3"})
    s = T.GpuSensors(bdfs[0], root)
    assert s.read() == {"sclk_MHz": 2100, "power_W": 750.0}
    assert T.GpuSensors("0000:ee:00.0", root).read() == {}
