"""GPU (one MI355X): an RCCL collective in flight next to the SANM block kernel.

The block / tile / fused streaming kernels need every workgroup of their grid resident and spin on each other's counters; an RCCL kernel parked on a
few CUs is the co-tenant that splits a cluster. The rule (include/asr_mi355x.h: asr_device_foreign_begin / _end; dist.foreign_section): a collective
waits for the cluster pass in flight, and while it is on the GPU every compute call takes its cluster-free path. Pinned here with a world-size-1
`nccl` (= RCCL) process group: a host thread issues broadcasts / all-reduces / gathers through dist.py on torch's stream while another drives batch
passes through the block kernel -- no cluster may give up, every pass must return the quiet run's tokens, and the gate's books must balance."""
import os
import threading
import time

import numpy as np
import pytest
import torch

from conftest import sub
from test_oracle_sensevoice import kaldi_audio, sensevoice_setup

pytestmark = pytest.mark.gpu
BF16 = 0


@pytest.fixture(scope="module")
def rccl_world1():
    import torch.distributed as dist
    if dist.is_initialized():
        yield dist
        return
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    store = f"tcp://127.0.0.1:{29600 + os.getpid() % 2000}"
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method=store, rank=0, world_size=1)
    yield dist
    dist.destroy_process_group()


def test_collectives_in_flight_never_meet_a_cluster_kernel(rccl_world1, capfd):
    dist = rccl_world1
    dp, eng = sub("dist"), sub("engine")
    dev = torch.device("cuda", 0)
    cfg, ck = sensevoice_setup("sensevoice_small")
    B, iters = 64, 40
    audios = [kaldi_audio(9100 + i, 128000) for i in range(B)]
    langs = [i % 7 for i in range(B)]
    sess = eng.SenseVoiceSession.from_checkpoint(cfg, ck, precision=BF16)
    ref = sess.run(audios, langs)
    sess.run(audios, langs)                                                    # second pass: the captured graph of the block path
    assert sess.sanm_stats() == {"giveups": 0, "cooldown": 0, "foreign_diverted": 0, "block_kernel": True}
    # a diverted pass takes the four-launch path: exactly the tokens of a session that never uses the block kernel (other K order than the block path:
    # on this random head near-tie frames differ between the two paths, so each path is compared with its own quiet run)
    os.environ["ASR_SANM_BLOCK"] = os.environ["ASR_SANM_TILES"] = "0"
    try:
        quiet4 = eng.SenseVoiceSession.from_checkpoint(cfg, ck, precision=BF16)
    finally:
        del os.environ["ASR_SANM_BLOCK"], os.environ["ASR_SANM_TILES"]
    ref4 = quiet4.run(audios, langs)
    del quiet4
    with dp.foreign_section(dev):
        inside = sess.run(audios, langs)
    st = sess.sanm_stats()
    assert st["foreign_diverted"] == 1 and st["giveups"] == 0
    assert all(np.array_equal(a, b) for a, b in zip(inside, ref4))
    capfd.readouterr()
    stop = threading.Event()
    n_coll = [0]
    payload = torch.arange(1 << 20, dtype=torch.float32, device=dev)           # 4 MB: a collective that lives for a while
    side = torch.cuda.Stream(device=dev)
    with dp.foreign_section(dev):                                              # (the first collective builds the communicator: seconds, not part of what is tested)
        with torch.cuda.stream(side):
            dist.all_reduce(payload)
            dist.broadcast(payload, 0)
    g0 = dp.foreign_stats(dev)

    def collectives():
        torch.cuda.set_device(0)
        while not stop.is_set():
            with dp.foreign_section(dev):
                with torch.cuda.stream(side):
                    dist.all_reduce(payload)
                    dist.broadcast(payload, 0)
            slab = dp.pack_hypotheses(np.ones((B, 8), np.int32), np.full(B, 8, np.int32), 8)
            got = dp.gather_hypotheses(slab, dev)                              # world 1: returns the slab itself, no section needed
            assert len(got) == 1
            dp.max_over_ranks(1.0, dev)
            n_coll[0] += 1
            time.sleep(0.002)

    bad, lat = [], []
    t = threading.Thread(target=collectives)
    t.start()
    try:
        it = -1
        while it + 1 < iters or (n_coll[0] < 10 and it < 20 * iters):           # at least `iters` passes, and until ten collective rounds ran beside them
            it += 1
            t0 = time.perf_counter()
            got = sess.run(audios, langs)
            lat.append(time.perf_counter() - t0)
            st_now = sess.sanm_stats()
            diverted = st_now["foreign_diverted"] != st["foreign_diverted"]
            st = st_now
            target = inside if diverted else ref                               # a diverted pass ran the four-launch path: its own (deterministic) tokens
            if not all(np.array_equal(a, b) for a, b in zip(got, target)):
                bad.append((it, diverted))
    finally:
        stop.set()
        t.join()
    err = capfd.readouterr().err
    g1 = dp.foreign_stats(dev)
    assert not bad, f"passes whose tokens differ from the quiet run of the same path: {bad[:5]}"
    assert "gave up" not in err and "redone" not in err, err[-500:]
    assert sess.sanm_stats()["giveups"] == 0 and sess.sanm_stats()["cooldown"] == 0
    assert n_coll[0] >= 5, n_coll                                               # the collectives really ran beside the passes
    d = {k: g1[k] - g0[k] for k in g0}
    assert d["sections"] >= n_coll[0]                                          # one section per bracketed group (+ those dist.py opens itself)
    assert d["cluster_passes_admitted"] + d["cluster_passes_diverted"] == it + 1
    print(f"{it + 1} block-kernel passes beside {n_coll[0]} collective rounds: {d['cluster_passes_diverted']} passes diverted to the cluster-free path, "
          f"{d['sections_that_waited']} sections waited for a pass; median pass {np.median(lat) * 1e3:.2f} ms, max {max(lat) * 1e3:.2f} ms")


def test_foreign_end_without_begin_is_an_error():
    lib = sub("_lib")
    h = lib.load()
    assert h.asr_device_foreign_begin(0) == 0 and h.asr_device_foreign_begin(0) == 0          # sections nest
    assert h.asr_device_foreign_end(0) == 0 and h.asr_device_foreign_end(0) == 0
    assert h.asr_device_foreign_end(0) != 0 and b"no foreign section" in h.asr_last_error()
