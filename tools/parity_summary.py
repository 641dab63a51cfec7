"""Condense a VCVITS_PARITY_STATS file (tests/golden_util.record_stats) into the summaries under profiles/:
python tools/parity_summary.py gpurun_out/parity_all.txt profiles/r5_parity_stats   (tools/suite_loop.sh records and condenses it)"""
import collections
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = collections.defaultdict(list)
for l in open(src):
    p = l.split()
    if len(p) < 3:
        continue
    kv = {}
    for t in p[2:]:
        k, _, v = t.partition("=")
        try:
            kv[k] = float(v)
        except ValueError:
            pass
    rows[p[0]].append((p[1], kv))


def table(f, title, items, key, n=25):
    f.write("\n# %s (worst %d by %s)\n" % (title, n, key))
    for name, kv in sorted(items, key=lambda r: -r[1].get(key, 0.0))[:n]:
        f.write("%-70s %s\n" % (name[:70], " ".join("%s=%.4g" % (k, v) for k, v in kv.items())))


with open(dst + "_f32.txt", "w") as f:
    k = rows.get("kinked", [])
    f.write("# close_kinked statistics of the full-width fp32 step tests (tests/test_full_width_step_gpu.py, test_dropout_step_gpu.py,\n"
            "# test_48k_gpu.py ...): %d tensors; produced by VCVITS_PARITY_STATS=... pytest -m gpu, condensed by tools/parity_summary.py\n" % len(k))
    for lo, hi, label in ((65536, 1e18, ">= 65536 elements"), (4096, 65536, "4096 .. 65535 elements"), (0, 4096, "< 4096 elements")):
        sel = [r for r in k if lo <= r[1].get("n", 0) < hi]
        if not sel:
            continue
        mx = lambda key: max(r[1].get(key, 0.0) for r in sel)
        f.write("%-24s %5d tensors: worst rel_l2 %.3g, beyond tol %.3g, beyond 4 tol %.3g, beyond 10 tol %.3g, max element / scale %.3g\n"
                % (label, len(sel), mx("rel_l2"), mx("frac_beyond_tol"), mx("frac_beyond_4tol"), mx("frac_beyond_10tol"), mx("max_err_over_scale")))
    for key in ("rel_l2", "frac_beyond_tol", "max_err_over_scale"):
        table(f, "tensors of >= 4096 elements", [r for r in k if r[1].get("n", 0) >= 4096], key)
with open(dst + "_bf16.txt", "w") as f:
    b = rows.get("bf16step", []) + rows.get("bf16wave", [])
    f.write("# bf16-mode whole-step statistics against the fp32 oracle (tests/test_bf16_step_gpu.py) and waveform RMS\n"
            "# (tests/test_bf16_gpu.py): losses, per-network totals, worst tensors; condensed by tools/parity_summary.py\n")
    for name, kv in b:
        if "/loss_" in name or "/TOTAL/" in name or "/WORST" in name or name.startswith("generator/"):
            f.write("%-50s %s\n" % (name, " ".join("%s=%.4g" % (k, v) for k, v in kv.items())))
    per = [r for r in b if "n" in r[1] and r[1].get("rel_l2", 0) < 1.0]
    table(f, "gradient tensors (analytically-zero ones excluded)", per, "rel_l2", 40)
if rows.get("x3") or rows.get("x3wgrad"):
    with open(dst.replace("parity_stats", "x3_vs_f64") + ".txt", "w") as f:
        f.write("# max-norm error / output scale against float64 torch CPU convolutions (tests/test_conv_x3_gpu.py):\n"
                "# x3_9 = split-operand kernel, nine product terms; x3_6 = six terms; pk = fp32-input MFMA kernel (fmaf chain)\n")
        for name, kv in rows.get("x3", []) + rows.get("x3wgrad", []):
            f.write("%-44s %s\n" % (name, " ".join("%s=%.3e" % (k, v) for k, v in kv.items())))
print(open(dst + "_f32.txt").read()[:1500])
