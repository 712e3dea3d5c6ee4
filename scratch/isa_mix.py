"""scratch/isa_mix.py <asm file> <kernel substring>...: static instruction mix of the kernels of a device assembly file
(scratch/kernel_regs.sh leaves one in /tmp/last_kernels.s): instruction counts by class."""
import collections, re, subprocess, sys
path, wants = sys.argv[1], sys.argv[2:]
txt = open(path).read()
for m in re.finditer(r"^((?:_Z|k_)\w+):[^\n]*\n(.*?)^\.Lfunc_end", txt, re.S | re.M):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"^void rt::\(anonymous namespace\)::", "", name); name = re.sub(r"\(.*", "", name)
    if wants and not any(w in name for w in wants): continue
    c = collections.Counter()
    for ln in m.group(2).splitlines():
        ln = ln.strip()
        if not ln or ln.startswith((";", ".", "//")) or ln.endswith(":"): continue
        op = ln.split()[0]
        if op.startswith("v_") and "f64" in op: k = "valu f64"
        elif op.startswith("v_pk_"): k = "valu packed"
        elif op.startswith("v_cndmask") or op.startswith("v_cmp"): k = "valu cmp/select"
        elif op.startswith("v_mov") or op.startswith("v_accvgpr"): k = "valu mov"
        elif op.startswith("v_"): k = "valu other"
        elif op.startswith("scratch_"): k = "scratch " + ("load" if "load" in op else "store")
        elif op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"): k = "vmem " + ("load" if "load" in op else "store" if "store" in op else "atomic")
        elif op.startswith("ds_"): k = "lds"
        elif op.startswith("s_waitcnt"): k = "s_waitcnt"
        elif op.startswith("s_cbranch") or op.startswith("s_branch"): k = "branch"
        elif op.startswith("s_"): k = "salu"
        else: k = "other"
        c[k] += 1
    tot = sum(c.values())
    print("%s: %d instructions" % (name, tot))
    for k, v in c.most_common(): print("    %-18s %6d  %4.1f %%" % (k, v, 100.0 * v / tot))
