"""GPU box: can several ranks of one RCCL communicator share the box's single GPU?  Starts WORLD processes, each makes a communicator
rank on device 0 and gathers synthetic tiles with rt_comm_gather_tiles; prints what happened.  (RCCL documents that it refuses
duplicate devices; this records what the installed version does.)  usage: python3 tools/rccl_ranks_one_gpu.py [world]"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def worker(rank, world, idfile, W, H):
    import numpy as np, torch
    from raytracinggpu_amd import _rccl
    if rank == 0:
        uid = _rccl.unique_id()
        with open(idfile + ".tmp", "wb") as f: f.write(uid)
        os.rename(idfile + ".tmp", idfile)
    else:
        for _ in range(600):
            if os.path.exists(idfile): break
            time.sleep(0.1)
        uid = open(idfile, "rb").read()
    torch.cuda.set_device(0)
    comm = _rccl.Comm(0, rank, world, uid)
    rows = [r for t in range(rank, (H + 7) // 8, world) for r in range(t * 8, min(H, t * 8 + 8))]
    full = (np.arange(H * W * 4, dtype=np.float32).reshape(H, W, 4) % 1000.0)
    tiles = torch.from_numpy(full[rows].copy()).cuda()
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") if rank == 0 else None
    torch.cuda.synchronize()
    comm.gather_tiles(tiles.data_ptr(), W, H, 16, frame.data_ptr() if rank == 0 else None)
    comm.sync()
    if rank == 0:
        ok = bool((frame.cpu().numpy() == full).all())
        print("rank 0: gathered frame equals the source:", ok, "bytes received", comm.last_bytes, flush=True)
    comm.close()

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], 640, 250)
        sys.exit(0)
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    idfile = os.path.join(tempfile.mkdtemp(), "id")
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(world), idfile], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    for r, p in enumerate(ps):
        try:
            out, _ = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill(); out, _ = p.communicate(); out += "\n[timeout]"
        print("== rank %d rc %s\n%s" % (r, p.returncode, "\n".join(l for l in out.splitlines() if "amdgpu.ids" not in l)[-1500:]))
