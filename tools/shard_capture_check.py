"""One-rank RCCL rehearsal of the CAPTURED range-sharded step (LeapFrogSimulator.capture_step): the same sharded code path
(nccl group, asynchronous all-gather, split force, fused kick) eager and as one hipGraph replay, from the same state.
Start it from torch.distributed.run BEFORE anything touches the GPU:
    NBD_FORCE_SHARDED=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
        --master-port P tools/shard_capture_check.py [n] [steps]
Prints one JSON line: captured?, bit-identical over `steps` steps?, host enqueue time per step before / after."""
import json, os, sys, time
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "nbody-deep-sim_amd"), ROOT):
    sys.path.insert(0, _p)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    import torch
    import torch.distributed as dist
    from galaxify import simulation
    from nbd.plummer import generate_plummer
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    os.environ["NBD_FORCE_SHARDED"] = "1"
    p, v, m = generate_plummer(n, seed=11)
    kw = dict(positions=p, velocities=v, masses=m, g_const=1.0, softening=0.1, dt=0.01, calc_energy=False, device="cuda",
              process_group=dist.group.WORLD)
    a, b = simulation.LeapFrogSimulator(**kw), simulation.LeapFrogSimulator(**kw)
    assert a._sharded and b._sharded
    warm = 3
    for _ in range(warm):
        a.step()                                  # capture_step() runs the same number of warm-up steps on b

    def enqueue_ms(sim, k=50):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k):
            sim.step()
        dt = (time.perf_counter() - t0) / k * 1e3
        torch.cuda.synchronize()
        return dt
    captured = b.capture_step(warmup=warm)
    same = True
    for i in range(steps):
        a.step(); b.step()
        torch.cuda.synchronize()
        same = same and all(torch.equal(getattr(a, k), getattr(b, k)) for k in ("positions", "velocities", "accelerations"))
    out = {"n": n, "steps": steps, "backend": dist.get_backend(), "captured": bool(captured), "bit_identical": bool(same),
           "host_enqueue_ms_eager": enqueue_ms(a), "host_enqueue_ms_captured": enqueue_ms(b) if captured else None,
           "phases_ms": a.step_phases()}
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
