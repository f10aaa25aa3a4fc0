"""Host draw worker of replay mode (run_many_replay): a plain child process that owns the two NumPy generators of a
fixed subset of chains and, on request, draws their next n proposals in the reference's order (draw_chunk:
MCMC.py:742-778, :176-254, :1253-1261, :1336), writing the masked fields into the parent's shared-memory buffer.

    python -m mcmc_gpu_amd._draw_worker <fd in> <fd out>

Protocol: length-prefixed pickles over two inherited pipes.  init {rf_param, H, W, update_in_region, region_mask,
shm_names, shm_shape, slots, rf_states, chain_states} -> 'ready'; ('draw', which, n) -> [(slot, size_idx, centre, u)];
('states',) -> {slot: (rf_state, chain_state)}; ('quit',).  Started directly (not through multiprocessing) so that the
workers start concurrently and never re-import the caller's __main__ (a torch import costs ~1 s per worker there).
Imports numpy only."""
import os
import pickle
import struct
import sys


def send(f, obj):
    b = pickle.dumps(obj, protocol=4)
    f.write(struct.pack('<Q', len(b)))
    f.write(b)
    f.flush()


def recv(f):
    h = f.read(8)
    if len(h) < 8:
        raise EOFError
    n, = struct.unpack('<Q', h)
    return pickle.loads(f.read(n))


def main(fd_in, fd_out):
    import numpy as np
    from multiprocessing import shared_memory
    from . import MCMC_gpu
    fin, fout = os.fdopen(fd_in, 'rb'), os.fdopen(fd_out, 'wb')
    init = recv(fin)
    rf_param = dict(init['rf_param'], rng_seed=0)
    shms = [shared_memory.SharedMemory(name=nm) for nm in init['shm_names']]
    try:    # attaching registered the segments with a resource tracker of this process: they are the parent's to unlink
        from multiprocessing import resource_tracker
        for m in shms:
            resource_tracker.unregister(m._name, 'shared_memory')
    except Exception:
        pass
    bufs = [np.ndarray(init['shm_shape'], dtype=np.float64, buffer=m.buf) for m in shms]
    H, W, upd, region = init['H'], init['W'], init['update_in_region'], init['region_mask']
    chains = {}
    for slot, st_rf, st_ch in zip(init['slots'], init['rf_states'], init['chain_states']):
        rf = MCMC_gpu.initiate_RF_by_instance(rf_param)
        rf.rng.bit_generator.state = st_rf
        rng = np.random.default_rng(0)
        rng.bit_generator.state = st_ch
        chains[slot] = (rf, rng)
    send(fout, 'ready')
    while True:
        msg = recv(fin)
        if msg[0] == 'quit':
            break
        if msg[0] == 'states':
            send(fout, {s: (rf.rng.bit_generator.state, rng.bit_generator.state) for s, (rf, rng) in chains.items()})
            continue
        _, which, n = msg
        out = []
        for slot, (rf, rng) in chains.items():
            si, ce, u, fields = MCMC_gpu.draw_chunk(rf, rng, n, H, W, upd, region)
            dst = bufs[which][slot]
            for s, f in enumerate(fields):
                dst[s, :f.size] = f.ravel()
            out.append((slot, si, ce, u))
        send(fout, out)
    for m in shms:
        m.close()


if __name__ == '__main__':
    main(int(sys.argv[1]), int(sys.argv[2]))
