"""The library's batch policy (include/frieda_hip.h "batch policy": frieda_workspace_bytes / frieda_batch_plan) and the NUMA placement
parser — host logic, no GPU.  The policy is what frieda_prove_many / frieda_commit_many apply per device and what bench.py's measured
loop asks for; the GPU side (results equal whatever the cut) is tests/test_gpu_parity.py::test_batch_policy_*."""
import ctypes as C

import pytest

from util import blob_len_for


@pytest.fixture(scope="module")
def fa():
    import frieda_amd

    return frieda_amd


def _cfg(fa, last=0, blowup=4):
    return fa.PcsConfig(fa.FriConfig(blowup, last, 20), 20)


def test_workspace_bytes_follow_the_domain(fa):
    ws = {n: fa.workspace_bytes(blob_len_for(n), 4) for n in (12, 16, 20, 22, 24)}
    assert all(v > 0 for v in ws.values())
    # eval 16 N + first tree 64 N + inner layers (16 + 64) N + coefficients: ~161 bytes per domain point
    assert 150 * (1 << 24) < ws[24] < 170 * (1 << 24)
    assert 3.9 < ws[24] / ws[22] < 4.1 and 3.9 < ws[22] / ws[20] < 4.1
    # commit keeps only the evaluation and the root scratch
    wc = fa.workspace_bytes(blob_len_for(24), 4, prove=False)
    assert 16 * (1 << 24) < wc < 0.2 * ws[24]
    # shapes the entry points refuse: domain above 2^28, last layer larger than the polynomial, blowup out of range
    assert fa.workspace_bytes(blob_len_for(28), 4) > 0 and fa.workspace_bytes(blob_len_for(29), 4) == 0
    assert fa.workspace_bytes(100, 4, 10) == 0 and fa.workspace_bytes(100, 29) == 0
    assert fa.workspace_bytes(0, 4, prove=False) > 0 and fa.workspace_bytes(0, 4, prove=True) == 0  # the empty blob commits, but is too small for FRI


@pytest.mark.parametrize("n,count,expect", [(24, 20, [10, 10]), (24, 60, [15] * 4), (24, 4, [2, 2]), (24, 7, [4, 3]), (24, 9, [5, 4]), (24, 11, [6, 5]),
                                            (24, 40, [10] * 4), (22, 64, [32, 32]), (22, 8, [4, 4]), (22, 200, [50] * 4), (20, 64, [32, 32]), (20, 256, [128, 128]),
                                            (11, 300, [150, 150])])
def test_plan_at_the_baseline_sizes(fa, n, count, expect):
    assert fa.batch_plan(blob_len_for(n), count, _cfg(fa)) == expect


def test_plan_invariants(fa):
    cfg = _cfg(fa)
    for n in (8, 12, 16, 20, 22, 24):
        ws = fa.workspace_bytes(blob_len_for(n), 4)
        budget = 16 * fa.workspace_bytes(blob_len_for(24), 4)
        for in_flight in (1, 2, 3):
            for count in list(range(0, 40)) + [63, 64, 65, 300, 1000, 4097]:
                cut = fa.batch_plan(blob_len_for(n), count, cfg, in_flight=in_flight)
                assert sum(cut) == count and all(c >= 1 for c in cut)
                if not cut:
                    continue
                assert max(cut) - min(cut) <= 1 and cut == sorted(cut, reverse=True)
                assert len(cut) % in_flight == 0 or len(cut) == count  # a multiple of the calls in flight (fewer blobs than contexts: one each)
                assert max(cut) == 1 or max(cut) * ws <= budget  # the budget is a ceiling
                if count >= in_flight:
                    assert len(cut) >= in_flight  # every context gets a call


def test_plan_caps_the_blobs_per_call(fa):
    cut = fa.batch_plan(1024, 100000, _cfg(fa))  # tiny blobs: the budget would allow 65535 per call
    assert sum(cut) == 100000 and max(cut) <= 4096 and len(cut) % 2 == 0 and max(cut) - min(cut) <= 1


def test_plan_for_commits_and_errors(fa):
    from frieda_amd import _lib

    L = _lib.lib()
    cut = fa.batch_plan(blob_len_for(24), 64, log_blowup_factor=4, prove=False)
    assert cut == [32, 32]  # commit workspaces are small: the spread rule (one call per context) binds
    n = C.c_uint32(7)
    out = (C.c_uint32 * 2)()
    assert L.frieda_batch_plan(None, 1000, 4, 0, 1, 10, 2, out, 1, C.byref(n)) == _lib.ERR_ARG and n.value == 2  # cap too small: the count is still reported
    assert L.frieda_batch_plan(None, 1000, 4, 0, 1, 10, 0, None, 0, C.byref(n)) == _lib.ERR_ARG  # in_flight 0
    assert L.frieda_batch_plan(None, 1000, 40, 0, 1, 10, 2, None, 0, C.byref(n)) == _lib.ERR_ARG  # blowup out of range
    assert L.frieda_batch_plan(None, 1000, 4, 0, 1, 10, 2, None, 0, None) == _lib.ERR_ARG


@pytest.mark.parametrize("text,expect", [("0-3,8,10-11\n", [0, 1, 2, 3, 8, 10, 11]), ("5", [5]), ("0-127", list(range(128))), ("", []), ("\n", []),
                                         ("0-1,64-65 \n", [0, 1, 64, 65])])
def test_cpulist_parser(text, expect):
    from frieda_amd import _lib

    L = _lib.lib()
    n = C.c_size_t(0)
    out = (C.c_int * 256)()
    assert L.frieda_test_parse_cpulist(text.encode(), out, 256, C.byref(n)) == _lib.OK
    assert [out[i] for i in range(n.value)] == expect


@pytest.mark.parametrize("text", ["a", "3-1", "1,,2", "1-", "-3", "1;2", "70000", "0-3,x"])
def test_cpulist_parser_rejects(text):
    from frieda_amd import _lib

    L = _lib.lib()
    n = C.c_size_t(0)
    out = (C.c_int * 16)()
    assert L.frieda_test_parse_cpulist(text.encode(), out, 16, C.byref(n)) == _lib.ERR_FORMAT
    assert L.frieda_test_parse_cpulist(b"0-31", out, 16, C.byref(n)) == _lib.ERR_ARG and n.value == 32


def test_numa_placement_on_a_two_socket_tree(tmp_path):
    """VERDICT r05 task 4c: on a node whose sysfs puts two GPUs on different NUMA nodes, their worker threads must get different CPU
    sets — each the CPUs of its own node, inside the process's affinity — and devices of one node the same set.  A fake sysfs tree
    plays the two-socket 8-GPU node (frieda_test_near_cpus runs the rule frieda_multi_create uses, multi.cpp cpus_near_bus_id)."""
    import os

    from frieda_amd import _lib

    L = _lib.lib()
    have = sorted(os.sched_getaffinity(0))
    if len(have) < 2:
        pytest.skip("needs two usable CPUs")
    half = len(have) // 2
    node_cpus = {0: have[:half], 1: have[half:]}
    for k, cpus in node_cpus.items():
        d = tmp_path / "devices" / "system" / "node" / f"node{k}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(",".join(str(c) for c in cpus) + ",9999\n")  # (a CPU outside the affinity mask must be dropped)
    gpus = {f"0000:{0x05 + 0x10 * i:02x}:00.0": (0 if i < 4 else 1) for i in range(8)}  # GPUs 0 - 3 on socket 0, 4 - 7 on socket 1
    for bus, node in gpus.items():
        d = tmp_path / "bus" / "pci" / "devices" / bus
        d.mkdir(parents=True)
        (d / "numa_node").write_text(f"{node}\n")
    (tmp_path / "bus" / "pci" / "devices" / "0000:ff:00.0").mkdir()
    (tmp_path / "bus" / "pci" / "devices" / "0000:ff:00.0" / "numa_node").write_text("-1\n")  # a platform that reports no affinity

    def near(bus):
        n = C.c_size_t(0)
        out = (C.c_int * 4096)()
        assert L.frieda_test_near_cpus(str(tmp_path).encode(), bus.encode(), out, 4096, C.byref(n)) == _lib.OK
        return [out[i] for i in range(n.value)]

    sets = {bus: near(bus.upper() if i % 2 else bus) for i, bus in enumerate(gpus)}  # (hipDeviceGetPCIBusId's upper-case hex is folded)
    for bus, node in gpus.items():
        assert sets[bus] == node_cpus[node], (bus, sets[bus])
    for a in gpus:
        for b in gpus:
            if gpus[a] != gpus[b]:
                assert not set(sets[a]) & set(sets[b]), "two workers of different NUMA nodes share CPUs"
    assert near("0000:ff:00.0") == [] and near("0000:ee:00.0") == []  # no affinity reported / unknown device: not pinned
