"""Multi-device context (gaml_hip_create_multi) on host-only shards: the partition logic, the forwarding of the C ABI
and the argument checks -- no GPU needed (device = -1 shards align and place windows, they cannot score)."""
import numpy as np
import pytest

from gaml_amd import synth


def _setup(n_pairs=900, G=30_000, seed=7):
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(900, 2500)))
    pr = synth.make_paired_reads(genome, n_pairs, 100, 250.0, 25.0, 0.01, seed)
    return g, (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))


def test_host_only_shards_partition_reads_and_windows(built):
    from gaml_amd import api
    g, reads = _setup()
    gb, go = g.packed()
    walk = synth.genome_walk(g)
    paths = [walk[:9], walk[9:]]
    whole = api.Context(device=-1)
    whole.set_graph(gb, go)
    rs = whole.add_paired(api.paired_cfg(250.0, 25.0), *reads)
    whole.debug_prepare(paths)

    multi = api.Context(devices=[-1, -1, -1])
    assert multi.num_shards() == 3 and whole.num_shards() == 1
    assert multi.exchange() == "host"  # no device, no communicator
    multi.set_graph(gb, go)
    assert multi.add_paired(api.paired_cfg(250.0, 25.0), *reads) == rs
    assert multi.num_readsets() == 1 and multi.readset_kind(rs) == 1 and multi.readset_reads(rs) == 900
    assert multi.num_nodes() == g.n_nodes and multi.node_len(2) == g.node_len(2)
    multi.debug_prepare(paths)
    # every shard registered the same windows; the union of the shards' records is the unsharded window
    assert multi.window_count(rs, 0) == whole.window_count(rs, 0) > 0
    seen = 0
    for mate in (0, 1):
        for wid in range(whole.window_count(rs, mate)):
            w = whole.debug_window_walk(rs, mate, wid)
            a, b = whole.window_records(rs, mate, w), multi.window_records(rs, mate, w)
            assert b is not None and np.array_equal(a, b), (mate, wid)
            seen += len(a)
    assert seen > 900
    # a window nobody cached
    assert multi.window_records(rs, 0, [walk[0], walk[1], walk[2], walk[3], walk[4], walk[5], walk[6]]) is None
    # scoring needs devices: the error names the shard
    with pytest.raises(api.GamlHipError) as e:
        multi.calc_prob(paths)
    assert e.value.code == api.ENODEVICE and "shard 0" in str(e.value)
    # what only makes sense for one shard per process is refused, not silently applied to shard 0
    with pytest.raises(api.GamlHipError) as e:
        multi.eval_begin(paths)
    assert e.value.code == api.ESTATE
    with pytest.raises(api.GamlHipError):
        multi.set_exchange("rccl")  # shards without their own GPU have no communicator
    multi.close()


def test_records_from_outside_are_validated(built):
    """gaml_hip_put_window_records refuses what the device tables cannot index: edit distance beyond the
    read / 255, orientation outside {0, 1}, positions outside the window, unknown reads."""
    from gaml_amd import api
    g, reads = _setup(n_pairs=50)
    ctx = api.Context(device=-1)
    ctx.set_graph(*g.packed())
    rs = ctx.add_paired(api.paired_cfg(250.0, 25.0), *reads)
    walk = synth.genome_walk(g)
    win = [walk[2], walk[3]]
    good = np.array([(5, 1, 3, 0)], api.ALIGMENT)
    for bad in [(5, 101, 3, 0), (5, 300, 3, 0), (5, -1, 3, 0), (5, 1, 3, 2), (5, 1, 3, -1), (-1, 1, 3, 0),
                (10_000_000, 1, 3, 0), (5, 1, 50, 0), (5, 1, -1, 0)]:
        with pytest.raises(api.GamlHipError) as e:
            ctx.put_window_records(rs, 0, win, np.array([bad], api.ALIGMENT))
        assert e.value.code == api.EINVAL, bad
        assert ctx.window_records(rs, 0, win) is None  # nothing was cached by the failed call
    with pytest.raises(api.GamlHipError):
        ctx.put_window_records(rs, 0, [10_000], good)  # node outside the graph
    ctx.put_window_records(rs, 0, win, good)
    assert ctx.window_records(rs, 0, win).tolist() == [[5, 1, 3, 0]]
    with pytest.raises(api.GamlHipError) as e:
        ctx.put_window_records(rs, 0, win, good)
    assert e.value.code == api.ESTATE  # already cached


def test_create_from_env(built, monkeypatch):
    import ctypes as C
    from gaml_amd import api
    L = api.lib()
    h = C.c_void_p()
    monkeypatch.setenv("GAML_HIP_DEVICES", "-1,-1")
    assert L.gaml_hip_create_from_env(C.byref(h)) == 0 and L.gaml_hip_num_shards(h) == 2
    L.gaml_hip_destroy(h)
    monkeypatch.setenv("GAML_HIP_DEVICES", "-1")
    assert L.gaml_hip_create_from_env(C.byref(h)) == 0 and L.gaml_hip_num_shards(h) == 1
    L.gaml_hip_destroy(h)
    monkeypatch.setenv("GAML_HIP_DEVICES", "0;1")
    assert L.gaml_hip_create_from_env(C.byref(h)) == api.EINVAL
