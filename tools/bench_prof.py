import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
import gradslam_amd as gs
from gradslam_amd import parallel
rank, world, local = parallel.init_from_env()
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
slam, world_map, prev, lives, K, raw = bench.build_workload(gs, dev, seed=0)
poses = []
with torch.no_grad():
    p0 = None
    for i in range(5):
        p0 = bench.one_step(gs, slam, world_map, prev, lives[i % 4], K)
    parallel.gather_poses(torch.cat([p0] * 50, 1), world)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    t0 = time.perf_counter()
    for i in range(50):
        poses.append(bench.one_step(gs, slam, world_map, prev, lives[i % 4], K))
    t1 = time.perf_counter()
    pr.disable()
    torch.cuda.synchronize()
print("enqueue ms", 1e3 * (t1 - t0))
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
