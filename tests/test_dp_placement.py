"""Host placement of the data-parallel ranks (fgnn_amd/dp.py: one rank = one GPU = its share of the NUMA-local cores) — the
pure planning logic, no device needed."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'factor-graph-neural-network_amd'))

from fgnn_amd import dp   # noqa: E402


def test_cpulist_round_trip():
    assert dp._cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    assert dp._cpulist('') == []
    assert dp._ranges([11, 0, 1, 2, 3, 8, 10]) == '0-3,8,10-11'
    assert dp._cpulist(dp._ranges(list(range(0, 64)) + list(range(128, 192)))) == list(range(0, 64)) + list(range(128, 192))


def test_ranks_of_one_numa_domain_get_disjoint_equal_shares():
    allowed = range(128)
    node0 = list(range(0, 64))
    shares = [dp.plan_rank_cpus(allowed, node0, slot, 4) for slot in range(4)]
    assert all(len(s) == 16 for s in shares)
    assert sorted(c for s in shares for c in s) == node0               # disjoint, covering, inside the domain


def test_without_numa_information_the_allowed_cores_are_split_by_rank():
    shares = [dp.plan_rank_cpus(range(30), [], r, 8) for r in range(8)]
    assert sorted(c for s in shares for c in s) == list(range(30))
    assert max(len(s) for s in shares) - min(len(s) for s in shares) <= 1


def test_a_cgroup_narrower_than_the_domain_and_fewer_cores_than_ranks():
    assert dp.plan_rank_cpus([4, 5, 6, 7], list(range(64)), 1, 2) == [6, 7]       # only what the process may use
    assert dp.plan_rank_cpus([70, 71], list(range(64)), 0, 2) == [70]              # domain not allowed at all: fall back to allowed
    assert [dp.plan_rank_cpus([0, 1], [], r, 8) for r in range(8)] == [[0], [1]] * 4    # shared round-robin, never empty
    assert dp.plan_rank_cpus([], [], 0, 8) == []


def test_topology_probe_is_quiet_without_kfd():
    assert isinstance(dp.xgmi_topology(), dict)
