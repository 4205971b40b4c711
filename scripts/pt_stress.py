"""Stress of several engines sweeping at once on one device (diagnostic): two identical worlds of W replicas (cfg-4 size), world A
sweeps with its W host threads running concurrently, world B one replica after the other; after every sweep the HS fields must be
identical and G equal to rounding.  usage: pt_stress.py [iterations] [W] [concurrent: 1 | 0]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from pt_twin import HostPT, ini_text, load_host
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
conc = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) != 0
L, U, nt, ns = 16, 8.0, 200, 10; n = L * L
betas = [8.0 - 0.4 * r for r in range(W)]; seeds = [1000 + r for r in range(W)]
h = load_host(); ini = ini_text(L, U, nt, ns)
A = HostPT(h, ini, betas, seeds); B = HostPT(h, ini, betas, seeds)
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def dump_event(it, r, a, b):
    """Everything one event can tell, into gpurun_out/pt_stress_event_<it>_<r>.npz + a text summary: the first differing (slice, site),
    the differing entries per slice, |dG| per 32 x 32 tile (the flush workgroups' tiles: a missed or stale flush shows as ONE hot
    tile, a stale prefetch of the walk as a hot row / column band), and for BOTH worlds' engines the per-stabilisation wrap errors and
    per-slice accepted counts of the last half sweep (a sweep call = forward + backward: the last half sweep is the backward one), the
    hand-off words of the persistent slice kernel, its launch count and the slice path."""
    n_stack = -(-nt // ns)
    da, db = A.debug(r, n_stack, nt), B.debug(r, n_stack, nt)
    diff = a["fields"] != b["fields"]
    sl, st = np.nonzero(diff)
    dG = np.abs(a["G"] - b["G"]).reshape(n // 32, 32, n // 32, 32).max(axis=(1, 3)) if n % 32 == 0 else np.abs(a["G"] - b["G"])
    os.makedirs(OUT, exist_ok=True)
    base = os.path.join(OUT, f"pt_stress_event_{it}_{r}")
    np.savez_compressed(base + ".npz", fields_A=a["fields"], fields_B=b["fields"], G_A=a["G"], G_B=b["G"], wrap_err_A=da["wrap_err"], wrap_err_B=db["wrap_err"],
                        accepted_A=da["accepted"], accepted_B=db["accepted"], arrive_A=da["arrive"], arrive_B=db["arrive"],
                        seq_A=da["seq"], seq_B=db["seq"], epoch_A=da["slice_epoch"], epoch_B=db["slice_epoch"])
    with open(base + ".txt", "w") as f:
        f.write(f"iteration {it} replica {r} (beta {betas[r]})\n")
        f.write(f"differing field entries (slice, site): {list(zip(sl.tolist(), st.tolist()))[:64]}\n")
        f.write(f"first differing slice {int(sl.min()) if len(sl) else None} = block {int(sl.min()) // ns if len(sl) else None}, position in block {int(sl.min()) % ns if len(sl) else None}\n")
        f.write("max|dG| per 32x32 tile (rows = row tile):\n" + np.array2string(dG, precision=2, max_line_width=200) + "\n")
        for name, d in (("A (concurrent)", da), ("B (sequential)", db)):
            f.write(f"world {name}: slice_path {d['slice_path']} slice_epoch {d['slice_epoch']} seq 0x{d['seq']:016x} (tag epoch {d['seq'] >> 40} window {(d['seq'] >> 32) & 0xff} "
                    f"flags/k 0x{d['seq'] & 0xffffffff:08x}) error {d['error']} solo_count {d['solo_count']}\n")
            f.write(f"  arrive[0..63] (tag = epoch << 8 | window): {[hex(int(x)) for x in d['arrive']]}\n")
            f.write(f"  wrap error per stabilisation of the last (backward) half sweep, in the order taken (block n_stack-1 first): {np.array2string(d['wrap_err'], precision=3)}\n")
            f.write(f"  accepted per slice of the last half sweep: {d['accepted'].tolist()}\n")
        bad_blocks = np.nonzero(da["wrap_err"] > 1e-4)[0]
        f.write(f"stabilisations of world A with wrap error > 1e-4 (index in the order taken): {bad_blocks.tolist()}\n")
        dacc = np.nonzero(da["accepted"] != db["accepted"])[0]
        f.write(f"slices whose accepted count differs between the worlds (last half sweep): {dacc.tolist()}\n")
    print(f"  event record written to {base}.txt / .npz", flush=True)


bad = 0; t0 = time.time()
for it in range(iters):
    A.sweeps(1, conc); B.sweeps(1, False)
    for r in range(W):
        a, b = A.get(r, n, nt), B.get(r, n, nt)
        nd = int((a["fields"] != b["fields"]).sum()); dG = float(np.abs(a["G"] - b["G"]).max())
        if nd or dG > 1e-7 * max(1.0, float(np.abs(b["G"]).max())):
            bad += 1
            sl = np.nonzero((a["fields"] != b["fields"]).any(axis=1))[0]
            per = (a["fields"] != b["fields"]).sum(axis=1)
            print(f"iteration {it} replica {r}: {nd} field entries differ in {len(sl)} slices, lowest {sl[:6]}, highest {sl[-6:]}, per slice (highest first) {per[sl][::-1][:12]}, "
                  f"max|dG| {dG:.3e}, max|G| {np.abs(b['G']).max():.2e}, max wrap err A {A.max_err(r):.3e} B {B.max_err(r):.3e}, accepted A/B {a['accepted']}/{b['accepted']}", flush=True)
            dump_event(it, r, a, b)
            A.set_fields(r, b["fields"])            # resynchronise and go on
    if it % 50 == 49:
        print(f"{it + 1} iterations, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("done:", iters, "iterations,", bad, "mismatches")
A.close(); B.close()
