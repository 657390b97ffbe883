"""Host batches sharded over a device set through the C ABI (include/labrador_ldpc_hip.h:
LABRADOR_LDPC_HIP_DEVICE_ALL / opts->devices; reference analogue perftest/src/main.rs:39-45).

CPU: the slice arithmetic the library uses, argument checking that needs no device.
GPU: a sharded call equals the single-device call and the oracle, whatever the device list; on a
one-GPU box the list repeats ordinal 0 (several host pipelines on one device), which exercises the same
splitting, worker threads and result placement; with more devices visible "all" shards for real."""
import ctypes

import numpy as np
import pytest

import labrador_ldpc_amd as la
from labrador_ldpc_amd import LDPCCode
from labrador_ldpc_amd.sharding import shard_range


def c_shard_range(batch, parts, index):
    first, count = ctypes.c_size_t(0), ctypes.c_size_t(0)
    st = la.lib.labrador_ldpc_hip_shard_range(batch, parts, index, ctypes.byref(first), ctypes.byref(count))
    return st, first.value, count.value


def test_shard_range_matches_the_python_harness():
    for batch in (0, 1, 5, 64, 1000003, 4194304, (1 << 33) + 7):
        for parts in (1, 2, 3, 8):
            got = [c_shard_range(batch, parts, i) for i in range(parts)]
            assert all(st == 0 for st, _, _ in got)
            assert [(f, c) for _, f, c in got] == [shard_range(batch, parts, i) for i in range(parts)]
            assert got[0][1] == 0 and sum(c for _, _, c in got) == batch
    assert c_shard_range(4194304, 8, 3) == (0, 3 * 524288, 524288)          # BASELINE config 4
    assert c_shard_range(10, 0, 0)[0] == -1 and c_shard_range(10, 2, 2)[0] == -1
    assert la.lib.labrador_ldpc_hip_shard_range(10, 2, 0, None, None) == -1


def test_device_set_arguments_are_checked_before_any_device_work():
    code = LDPCCode.TC128
    llrs = np.zeros((4, code.n()), dtype=np.float32)
    with pytest.raises(ValueError):
        code.decode_ms_batch(llrs, 5, devices=[])
    with pytest.raises(ValueError):
        code.decode_ms_batch(llrs, 5, devices="some")
    with pytest.raises(ValueError):
        code.decode_ms_batch(llrs, 5, devices=[0], stream=1234)
    out = np.zeros((4, code.output_len()), dtype=np.uint8)
    it, ok = np.zeros(4, dtype=np.uint32), np.zeros(4, dtype=np.uint8)
    args = (int(code), llrs.ctypes.data, out.ctypes.data, it.ctypes.data, ok.ctypes.data, 4, 5)
    # device memory cannot be sharded; a stream cannot be given; n_devices without a list; bad ordinal
    for opts, text in ((la.HipOpts(la.DEVICE_ALL, la.MEM_DEVICE, None, 0, 0, None), "MEM_HOST"),
                       (la.HipOpts(la.DEVICE_ALL, la.MEM_HOST, 1234, 0, 0, None), "stream"),
                       (la.HipOpts(-3, la.MEM_HOST, None, 0, 0, None), "opts->device"),
                       (la.HipOpts(0, la.MEM_HOST, None, 0, -1, None), "negative")):
        st = la.lib.labrador_ldpc_decode_ms_batch_f32(*args, ctypes.byref(opts))
        assert st == -1 and text in la.last_error(), la.last_error()


def test_result_buffers_are_validated():
    """preallocated buffers go to the C ABI as raw pointers, so shape/dtype/contiguity are checked."""
    code = LDPCCode.TC128
    llrs = np.zeros((4, code.n()), dtype=np.float32)
    good_out = np.zeros((4, code.output_len()), dtype=np.uint8)
    for kw in (dict(output=np.zeros((3, code.output_len()), dtype=np.uint8)),
               dict(output=np.zeros((4, code.output_len()), dtype=np.int8)),
               dict(output=np.zeros((4, 2 * code.output_len()), dtype=np.uint8)[:, ::2]),
               dict(iters=np.zeros(4, dtype=np.uint64)),
               dict(iters=np.zeros(5, dtype=np.uint32)),
               dict(success=np.zeros(4, dtype=np.int32)),
               dict(success=[0, 0, 0, 0])):
        with pytest.raises(ValueError):
            code.decode_ms_batch(llrs, 5, **kw)
    ro = good_out.copy()
    ro.flags.writeable = False
    with pytest.raises(ValueError):
        code.decode_ms_batch(llrs, 5, output=ro)
    with pytest.raises(ValueError):
        code.decode_ms_batch([[0.0] * code.n()], 5)
    with pytest.raises(la.LdpcHipError):
        code.decode_ms_batch(np.zeros((4, code.n()), dtype=np.uint16), 5)
    with pytest.raises(ValueError):
        code.encode_batch(np.zeros((2, code.k() // 8), dtype=np.uint8), codewords=np.zeros((2, 3), dtype=np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("name,dtype,frames", [("TM8192", np.float32, 301), ("TM2048", np.int8, 1000),
                                               ("TC128", np.float32, 5), ("TM5120", np.int16, 257)])
def test_sharded_host_batch_equals_single_device_and_oracle(name, dtype, frames):
    import oracle
    code = LDPCCode[name]
    rng = np.random.default_rng(77)
    llrs, _ = oracle.awgn_llrs(code, rng, frames, 2.5 if name != "TC128" else 4.0, dtype)
    want = oracle.decode_ms_batch(code, llrs, 20)[:3]
    single = code.decode_ms_batch(llrs, 20)
    ndev = la.device_count()
    lists = [[0], [0, 0], [0, 0, 0], "all", list(range(ndev)) * 2]
    if frames < 8:
        lists.append([0] * 8)                                     # more parts than frames: empty slices
    for devs in lists:
        got = code.decode_ms_batch(llrs, 20, devices=devs)
        for g, s, w in zip(got, single, want):
            assert (g == s).all() and (g == w).all(), f"{name} devices={devs}"


@pytest.mark.gpu
def test_sharded_large_batch_runs_the_chunked_pipelines_concurrently(monkeypatch):
    """Each slice is big enough to take the multi-chunk copy/kernel/copy pipeline of its worker."""
    import oracle
    monkeypatch.setenv("LABRADOR_LDPC_HIP_CHUNK", "512")
    code = LDPCCode.TM2048
    rng = np.random.default_rng(5)
    base, _ = oracle.awgn_llrs(code, rng, 1024, 2.0, np.float32)
    llrs = np.tile(base, (5, 1))[: 4099]
    want = oracle.decode_ms_batch(code, base, 25)[:3]
    for devs in ([0, 0], "all", [0, 0, 0]):
        out, it, ok = code.decode_ms_batch(llrs, 25, devices=devs)
        idx = np.arange(4099) % 1024
        assert (out == want[0][idx]).all() and (it == want[1][idx]).all() and (ok == want[2][idx]).all()


@pytest.mark.gpu
def test_sharded_encode_and_decode_bf():
    import oracle
    code = LDPCCode.TM1280
    rng = np.random.default_rng(9)
    data = rng.integers(0, 256, size=(333, code.k() // 8), dtype=np.uint8)
    cws = code.encode_batch(data, devices=[0, 0, 0])
    assert (cws == code.encode_batch(data)).all()
    assert all((cws[f] == oracle.copy_encode(code, data[f])).all() for f in range(0, 333, 37))
    rx = cws.copy()
    rx[:, 3] ^= 0x40
    out, it, ok = code.decode_bf_batch(rx, 30, devices=[0, 0])
    out1, it1, ok1 = code.decode_bf_batch(rx, 30)
    assert (out == out1).all() and (it == it1).all() and (ok == ok1).all() and ok.all()


@pytest.mark.gpu
def test_bad_ordinal_in_device_list_is_reported():
    code = LDPCCode.TC128
    llrs = np.zeros((4, code.n()), dtype=np.float32)
    with pytest.raises(la.LdpcHipError) as e:
        code.decode_ms_batch(llrs, 5, devices=[0, 99])
    assert "out of range" in str(e.value)


@pytest.mark.gpu
def test_concurrent_sharded_calls_from_several_host_threads():
    """The library's device workers are shared by every caller: concurrent sharded calls queue on them and must
    not mix their slices, statuses or error strings (include/labrador_ldpc_hip.h: 'functions are re-entrant')."""
    import threading
    import oracle
    code = LDPCCode.TM1536
    rng = np.random.default_rng(31)
    jobs = []
    for i in range(6):
        llrs, _ = oracle.awgn_llrs(code, rng, 500 + 37 * i, 2.5, np.float32 if i % 2 else np.int8)
        jobs.append((llrs, oracle.decode_ms_batch(code, llrs, 20)[:3], [[0, 0], "all", [0, 0, 0]][i % 3]))
    results, errors = [None] * len(jobs), []

    def run(i):
        try:
            for _ in range(3):
                results[i] = code.decode_ms_batch(jobs[i][0], 20, devices=jobs[i][2])
        except Exception as e:           # noqa: BLE001
            errors.append((i, e))

    threads = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    for (llrs, want, devs), got in zip(jobs, results):
        assert all((g == w).all() for g, w in zip(got, want)), devs


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.int8, np.float64])
def test_device_resident_parts_through_the_multi_call(dtype):
    """labrador_ldpc_decode_ms_batch_*_multi through the Python binding (LDPCCode.decode_ms_batch_multi): device-resident parts of
    unequal sizes, one of them empty, every one on the box's GPU(s) by ordinal -- each part's results equal the oracle's, and the call
    has returned only when they are in place (no synchronisation before the copies to the host)."""
    import oracle
    torch = pytest.importorskip("torch")
    code = LDPCCode.TM1536
    rng = np.random.default_rng(77)
    ndev = la.device_count()
    sizes = [301, 0, 1024, 77]
    parts, want = [], []
    for i, f in enumerate(sizes):
        llrs, _ = oracle.awgn_llrs(code, rng, max(f, 1), 2.5, dtype)
        llrs = llrs[:f]
        want.append(oracle.decode_ms_batch(code, llrs, 20)[:3] if f else None)
        parts.append(torch.from_numpy(llrs).to(torch.device("cuda", i % ndev)))
    for _ in range(2):
        got = code.decode_ms_batch_multi(parts, 20)
        for g, w, f in zip(got, want, sizes):
            assert g[0].shape[0] == f
            if f:
                assert all((x.cpu().numpy() == y).all() for x, y in zip(g, w))
    with pytest.raises(ValueError):
        code.decode_ms_batch_multi([parts[0], parts[2].to(torch.float16)], 20)
