#!/usr/bin/env python
"""Per-launch roofline table of ONE denoiser step of the antibody model at B = 256 (one lane), from a dispatch sequence
written by scripts/x3_seq.sh (rocpd_summary.py --sequence):

    python scripts/launch_budget.py gpurun_out/seq_x3/sequence.txt x3          > profiles/r03/ab256_x3_launch_budget.txt
    python scripts/launch_budget.py gpurun_out/seq_f32/sequence.txt default     > profiles/r03/ab256_launch_budget.txt
    python scripts/launch_budget.py gpurun_out/seq_allf32/sequence.txt allfp32  > profiles/r03/ab256_allfp32_launch_budget.txt

Routes (precision routes of include/hudiff_hip.h): `default` = f32_gemm (fp32 MFMA GEMMs + split-precision attention core, priced at 2500 / 3),
`allfp32` = f32_all, `x3` = split (the library default since round 4).

Every GEMM / attention launch is named by its place in the network (the launch order is fixed), priced with its algorithmic
FLOPs (2 M N K taps; attention 4 L^2 x 64 per head) and with the FEWEST HBM bytes its operands allow (activation operand read
once, every output written once, residual read once; weights stay in L2 / Infinity Cache), and compared with both ceilings of
its route: the matrix peak (fp32 MFMA 157.3 TFLOP/s; split route 2500 / 3) and the HBM streaming rate the guide measures as
achievable, 6.29 TB/s of the 8.0 TB/s spec (MI355X_MICROARCH.md "HBM"; rounds 2-3 priced 4.5 TB/s here, which made every
HBM-bound launch look closer to its floor than it is: VERDICT r3).  `floor` is max(FLOPs / peak, bytes / 6.29 TB/s): what a perfect
kernel of this shape would take; the fraction of the 8.0 TB/s spec is printed beside it; the last line sums the floors.
"""
import re
import sys

M = 256 * 291                      # activation rows of the batch
L, H = 291, 8
d, dh, D, Dh, A, Fd = 256, 128, 768, 384, 512, 256
HBM = 6.29e12                      # achievable HBM streaming rate of the guide (MI355X_MICROARCH.md); HBM_SPEC is the 8.0 TB/s figure
HBM_SPEC = 8.0e12


def gemm(K, N, taps=1):
    return 2.0 * M * N * K * taps


def main():
    path, route = sys.argv[1], sys.argv[2]
    x3 = route == "x3"
    peak = 2500e12 / 3 if x3 else 157.3e12
    att_peak = 157.3e12 if route == "allfp32" else 2500e12 / 3
    rows = []
    for ln in open(path).read().splitlines()[1:]:
        m = re.match(r"\s*([\d.]+)\s+([\d.]+) us\s+grid\s+(\d+)\s+(.*)", ln)
        if m:
            rows.append((float(m.group(2)), int(m.group(3)), m.group(4).strip()))
    # name the launches by position: 6 encoder blocks, 6 dual-conv blocks, attention blocks (the last one pruned)
    out = []
    gem = [r for r in rows if "gemm" in r[2] and r[1] >= 500 or "attn_x3_k" in r[2] or "attn_k<" in r[2] or "qkv_attn_x3_k" in r[2]]
    fused = any("qkv_attn_x3_k" in r[2] for r in rows)        # round 5: the Q|K|V projection runs inside the attention kernel (one launch)
    names = []
    for n in range(6):
        names += [("enc PFF1 256->128", gemm(d, dh), (d + dh)), ("enc tap GEMM 7x128->128", gemm(dh, dh, 7), (dh + dh)),
                  ("enc PFF3 128->256 (+x)", gemm(dh, d), (dh + d + d + (d if x3 and n < 5 else 0)))]
    for n in range(6):
        names += [("PFF1 768->384", gemm(D, Dh), (D + Dh)), ("tap GEMM 7x384->384", gemm(Dh, Dh, 7), (Dh + Dh)),
                  ("PFF3 384->768 (+x)", gemm(Dh, D), (Dh + D + D + (D if x3 else 0)))]
    att = 4.0 * L * L * 64 * H * 256
    blk = []
    def qkv_att(tag):
        # fused: the layer input is read once, O written once (Q round trip: one third of Q|K|V written and read back, L2-resident)
        if fused:
            return [(f"Q|K|V{tag} + attention core (fused)", gemm(D, 3 * A) + att, (D + A + 2 * A))]
        return [(f"Q|K|V 768->1536{tag}", gemm(D, 3 * A), (D + 3 * A)), ("attention core", att, (3 * A + A))]
    for n in range(5):
        last = n == 4
        blk += qkv_att("") + [("out-projection 512->768 (+x)", gemm(A, D), (A + D + D + (D if x3 else 0)))]
        if last:
            blk += [("K of the pruned attention 768->512", gemm(D, A), (D + A))]
            break
        blk += qkv_att(" (LN folded)") + [("out-projection 512->768 (+x)", gemm(A, D), (A + D + D + (D if x3 else 0))),
                ("FF1 768->256", gemm(D, Fd), (D + Fd)), ("FF2 256->768 (+x)", gemm(Fd, D), (Fd + D + D + (D if x3 else 0)))]
    names += blk
    if len(names) != len(gem):
        print(f"launch count mismatch: {len(names)} named vs {len(gem)} GEMM / attention dispatches", file=sys.stderr)
        sys.exit(1)
    agg = {}
    order = []
    for (us, grid, kname), (nm, fl, words) in zip(gem, names):
        a = agg.setdefault(nm, [0, 0.0, fl, words * M * 4.0])
        if a[0] == 0:
            order.append(nm)
        a[0] += 1
        a[1] += us
    big = sum(r[0] for r in gem)
    total = sum(r[0] for r in rows)
    print(f"one denoiser step, HuDiff-Ab, 256 rows, one lane, route {route}: {total / 1e3:.2f} ms of kernels in {len(rows)} dispatches; "
          f"GEMMs + attention cores {big / 1e3:.2f} ms, everything else {(total - big) / 1e3:.2f} ms")
    print(f"{'launch':38s} {'n':>3s} {'avg us':>8s} {'GFLOP':>7s} {'TFLOP/s':>8s} {'of peak':>8s} {'min MB':>7s} {'TB/s':>6s} {'of 6.29':>7s} {'of 8.0':>7s} {'bound':>6s} {'floor us':>9s}")
    floor_sum = 0.0
    for nm in order:
        n, us, fl, by = agg[nm]
        t = us / n * 1e-6
        tf, bw = fl / t, by / t
        pk = att_peak if nm == "attention core" else peak
        f_m, f_h = fl / pk, by / HBM
        floor = max(f_m, f_h)
        floor_sum += floor * n
        print(f"{nm:38s} {n:3d} {us / n:8.1f} {fl / 1e9:7.1f} {tf / 1e12:8.1f} {tf / pk:8.3f} {by / 1e6:7.0f} {bw / 1e12:6.2f} {bw / HBM:7.3f} {bw / HBM_SPEC:7.3f} "
              f"{'MFMA' if f_m >= f_h else 'HBM':>6s} {floor * 1e6:9.1f}")
    print(f"sum of floors of the launches above: {floor_sum * 1e3:.2f} ms against {big / 1e3:.2f} ms measured "
          f"({floor_sum * 1e3 / (big / 1e3):.2f} of it); with the other kernels unchanged the step would take "
          f"{floor_sum * 1e3 + (total - big) / 1e3:.2f} ms instead of {total / 1e3:.2f}")


if __name__ == "__main__":
    main()
