#!/usr/bin/env python3
"""Diagnostic only: phase stamps of the LAST association launch of a long streamed PointFusion run, i.e. with
the ICP target grown to the downsampled active map (needs make -C gradslam_amd/csrc diag).

Stamp slots per wave (16 x 8 bytes per wave, GS_STAMP / GS_COUNT in icp.hip; times are s_memrealtime ticks of 10 ns):
  6 kernel entry | 8 first batch of requests arrived | 9 row sums handed over (wave 0: sums ready) | 7 prologue barrier passed
  wave 0 (the folded step): 13 decision taken + state updated | 14 6x6 system solved | 10 step done
  wave 1 (planner): 13 window centres + majority displacement known | 14 plan published | 15 its share staged
  waves 2.. : 14 plan seen | 15 staged (wave 2, the lanes' wave: rows packed)
  0 / 1 / 2 / 3 search start / window scan done / proof + fallbacks done / end ; 5 (wave 0) HW_ID | XCC_ID: which CU
Every stamp costs ~0.1 us: compare runs of the same build, not absolute values with the product's timings.
Printed: the phases' percentiles, the same split by how many blocks share a CU, per role, and the old chunk-box figures."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gradslam_amd import _native
_native.LIB_PATH = os.path.join(ROOT, "gradslam_amd", "libgradslam_hip_diag.so")
import gradslam_amd as gs
from gradslam_amd.synthetic import make_sequence_cached as make_sequence

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
numiters = int(sys.argv[2]) if len(sys.argv) > 2 else 10  # < 10: the LAST frame only is cut short, so the last launch is association numiters + 1
dev = "cuda:0"
c, d, K, P = make_sequence(1, n, 480, 640, seed=100)
frames = gs.RGBDImages(c.to(dev), d.to(dev), K.to(dev), P.to(dev))
lib = _native.lib()
lib.gs_diag_set_buffer.argtypes = [ctypes.c_void_p]
nblk = 2048  # >= the association's grid (mixed 64 / 16-point tiling: 432 blocks at 19 200 candidates)
dbg = torch.zeros(nblk * 16 * 16, dtype=torch.int64, device=dev)
assert lib.gs_diag_set_buffer(dbg.data_ptr()) == 0
slam = gs.slam.PointFusion(odom="icp", dsratio=4, numiters=10, device=dev)
with torch.no_grad():
    if numiters == 10:
        pcs, poses = slam(frames)
    else:
        head = gs.RGBDImages(c[:, :n - 1].to(dev), d[:, :n - 1].to(dev), K.to(dev), P[:, :n - 1].to(dev))
        pcs, poses = slam(head)
        short = gs.slam.PointFusion(odom="icp", dsratio=4, numiters=numiters, device=dev)
        dbg.zero_()
        live = gs.RGBDImages(c[:, n - 1:n].to(dev), d[:, n - 1:n].to(dev), K.to(dev))
        prevf = gs.RGBDImages(c[:, n - 2:n - 1].to(dev), d[:, n - 2:n - 1].to(dev), K.to(dev), poses[:, n - 2:n - 1].contiguous())
        short._localize(pcs, live, prevf)
torch.cuda.synchronize()
print("frames", n, "map", int(pcs.num_points_per_pointcloud.item()))
raw = dbg.cpu().numpy().reshape(nblk, 16, 16)
a = raw.astype(np.float64)
live = a[..., 3].max(1) > 0
a = a[live]
place = raw[live][:, 0, 5]          # wave 0's slot 5: survivors | HW_ID << 16 | XCC_ID << 48
hw, xcc = (place >> 16) & 0xffffffff, (place >> 48) & 0xf
cu, sh_, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7
where = xcc * 4096 + se * 256 + sh_ * 16 + cu
a[..., 5] = (raw[live][..., 5] & 0xffff)
print("blocks with stamps", int(live.sum()))
tick = 1e-2
t0, t1, t2, t3, ns, t6, t7 = a[..., 0], a[..., 1], a[..., 2], a[..., 3], a[..., 4], a[..., 6], a[..., 7]
pc = lambda x, q: tuple(np.percentile(x, q))
print("prologue (folded step) us per wave p50 %.2f p99 %.2f" % pc((t7 - t6) * tick, [50, 99]))
print("prologue end -> search start p50 %.2f" % np.percentile((t0 - t7) * tick, 50))
w0, ws = a[:, 0, :], a[:, 1:, :]  # wave 0 = the folded step; waves 1.. = the staging path of the grid search
if (ws[..., 15] > 0).any():
    rel = lambda slot: ((ws[..., slot] - ws[..., 6]) * tick)[ws[..., slot] > 0]
    print("staging waves, us after kernel entry (p50): centre + tile displacement known %.2f | bands laid out %.2f | staged to LDS %.2f | barrier passed %.2f ; wave 0 (step) reaches the barrier at %.2f"
          % (np.percentile(rel(13), 50), np.percentile(rel(14), 50), np.percentile(rel(15), 50), np.percentile(rel(7), 50), np.percentile((w0[..., 7] - w0[..., 6]) * tick, 50)))
print("seed phase us  (per wave)  p50 %.2f p99 %.2f" % pc((t1 - t0) * tick, [50, 99]))
print("main loop us   (per wave)  p50 %.2f p99 %.2f max %.2f" % pc((t2 - t1) * tick, [50, 99, 100]))
print("barrier wait us(per wave)  p50 %.2f p99 %.2f" % pc((t3 - t2) * tick, [50, 99]))
print("block total us             p50 %.2f p99 %.2f max %.2f" % pc((t3.max(1) - t6.min(1)) * tick, [50, 99, 100]))
print("kernel entry spread us p50 %.2f p99 %.2f" % pc((t6.min(1) - t6.min()) * tick, [50, 99]))
print("kernel span us %.1f" % ((t3.max() - t6.min()) * tick))
need = a[:, 0, 12]
print("grid search: lanes needing the exact search per tile: mean %.2f, tiles with none %d of %d, max %d" % (need.mean(), int((need == 0).sum()), need.shape[0], int(need.max())))
why = raw[live][:, 0, 13]
rad = raw[live][:, 0, 14]
worst = int(np.argmax(need))
f = lambda w, sh: int((w >> sh) & 0xff)
print("tile with most lanes in need (%d): not fully staged %d | smaller radius %d | no certificate %d | moved beyond radius %d | best outside reach %d ; radius %d ; "
      "tile means (mm): sqrt(m) %.1f moved %.1f sqrt(bd) %.1f" % (need[worst], f(why[worst], 0), f(why[worst], 8), f(why[worst], 16), f(why[worst], 24), f(why[worst], 32),
      f(why[worst], 40), (rad[worst] & 0xfffff) / 1e3, ((rad[worst] >> 20) & 0xfffff) / 1e3, ((rad[worst] >> 40) & 0xfffff) / 1e3))
print("all tiles: lanes per reason (sum): not staged %d | radius %d | no certificate %d | moved %d | reach %d" % tuple(int(((why >> sh) & 0xff).sum()) for sh in (0, 8, 16, 24, 32)))
w15 = raw[live][:, 0, 15]
sel = w15 != 0
if sel.any():
    print("fresh certificates (tiles that ran the tile-level search: %d): lanes whose radius is the TILE-level bound: mean %.1f per tile ; "
          "sqrt(m_tile) mean %.1f mm ; mean per-lane sqrt(mm) %.1f mm" % (int(sel.sum()), (w15[sel] & 0xff).mean(), ((w15[sel] >> 8) & 0xffffff).mean() / 1e3,
          ((w15[sel] >> 32) & 0xffffff).mean() / 1e3))
print("coarse us per wave p50 %.2f p99 %.2f | fine us p50 %.2f p99 %.2f | barrier waits us p50 %.2f p99 %.2f | survivors tested per wave p50 %.0f" % (
    *pc(a[..., 8] * tick, [50, 99]), *pc(a[..., 9] * tick, [50, 99]), *pc(a[..., 10] * tick, [50, 99]), np.percentile(a[..., 11], 50)))
print("chunks scanned per block: mean %.1f max %d ; coarse survivors per block mean %.1f max %d" % (
    ns.sum(1).mean(), ns.sum(1).max(), a[..., 5].max(1).mean(), a[..., 5].max(1).max()))
ml = ((t2 - t1) * tick).max(1)
order = np.argsort(-ml)
print("slowest blocks: (main loop us, coarse survivors of the last round, chunks scanned)")
for b in order[:12]:
    print("   %.1f us  survivors %d  scanned %d" % (ml[b], a[b, :, 5].max(), ns[b].sum()))
print("median blocks:")
for b in order[len(order) // 2 - 3: len(order) // 2 + 3]:
    print("   %.1f us  survivors %d  scanned %d" % (ml[b], a[b, :, 5].max(), ns[b].sum()))
print("corr(main loop, survivors) %.2f  corr(main loop, scanned) %.2f" % (np.corrcoef(ml, a[..., 5].max(1))[0, 1], np.corrcoef(ml, ns.sum(1))[0, 1]))

uniq, counts = np.unique(where, return_counts=True)
print("distinct CUs used %d ; blocks per CU histogram %s" % (len(uniq), dict(zip(*np.unique(counts, return_counts=True)))))
per_cu = dict(zip(uniq, counts))
share = np.array([per_cu[w] for w in where])
bt = (t3.max(1) - t6.min(1)) * tick
pro = ((t7 - t6) * tick).max(1)
for k in sorted(set(share)):
    m = share == k
    print("  blocks on a CU hosting %d block(s): n=%d main loop p50 %.1f us max %.1f us | prologue p50 %.2f max %.2f | block total p50 %.2f max %.2f" % (
        k, m.sum(), np.percentile(ml[m], 50), ml[m].max(), np.percentile(pro[m], 50), pro[m].max(), np.percentile(bt[m], 50), bt[m].max()))
if (ws[..., 15] > 0).any():
    for k in sorted(set(share)):
        m = share == k
        relk = lambda slot: np.percentile(((ws[..., slot] - ws[..., 6]) * tick)[m][ws[..., slot][m] > 0], 50)
        seed_k, main_k = ((t1 - t0) * tick)[m], ((t2 - t1) * tick)[m]
        print("  %d block(s) on the CU: staging p50 centre %.2f | bands %.2f | LDS %.2f | barrier %.2f ; wave 0 at the barrier %.2f ; search start %.2f ; seed %.2f main %.2f ; end %.2f" % (
            k, relk(13), relk(14), relk(15), relk(7), np.percentile(((w0[..., 7] - w0[..., 6]) * tick)[m], 50),
            np.percentile(((t0 - t6) * tick)[m], 50), np.percentile(seed_k, 50), np.percentile(main_k, 50), np.percentile(((t3 - t6) * tick)[m], 50)))
if (a[..., 9] > 0).any():
    for k in sorted(set(share)):
        m = share == k
        r = lambda arr: np.percentile(((arr - a[..., 6]) * tick)[m], 50)
        w0r = lambda slot: np.percentile(((w0[..., slot] - w0[..., 6]) * tick)[m], 50)
        print("  %d block(s) on the CU: first batch arrived %.2f | rows summed %.2f | wave 0: sums ready %.2f, decision + state %.2f, solved %.2f, step done %.2f (us after kernel entry, p50)" % (
            k, r(a[..., 8]), r(a[..., 9]), w0r(9), w0r(13), w0r(14), w0r(10)))
if (a[..., 15] > 0).any() and a.shape[1] > 3:
    solo = share == min(set(share))
    wrel = lambda w, slot: np.percentile(((a[:, w, slot] - a[:, w, 6]) * tick)[solo & (a[:, w, slot] > 0)], 50)
    print("  per role (blocks alone on their CU, p50 us after kernel entry): wave 1 (planner) centre %.2f plan published %.2f staged %.2f | wave 2 (lanes) plan seen %.2f rows packed + staged %.2f | wave 3 (stager) plan seen %.2f staged %.2f | barrier passed %.2f" % (
        wrel(1, 13), wrel(1, 14), wrel(1, 15), wrel(2, 14), wrel(2, 15), wrel(3, 14), wrel(3, 15), wrel(3, 7)))
print("blocks per XCC:", dict(zip(*np.unique(xcc, return_counts=True))))
