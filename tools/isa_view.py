"""Compressed view of a kernel's memory/control instructions in a hipcc -save-temps .s file (dev helper).
python tools/isa_view.py <file.s> <kernel-name-substring> [first-line last-line]"""
import re, sys
src, name = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(name) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
pat = re.compile(r"s_barrier|s_waitcnt|global_load|buffer_load|flat_load|global_atomic|global_store|ds_write|ds_read|ds_add|s_cbranch|v_mfma|^\.LBB|scratch_")
prev, cnt, first = None, 0, 0
def flush():
    if prev is not None: print(f"{first:6d} x{cnt:<3d} {prev}")
for i in range(start, end + 1):
    l = lines[i].strip()
    if not pat.search(l): continue
    tok = l.split()
    key = tok[0] if not tok[0].startswith("s_waitcnt") and not tok[0].startswith("s_cbranch") else " ".join(tok[:2])
    if key == prev: cnt += 1
    else:
        flush(); prev, cnt, first = key, 1, i - start
flush()
