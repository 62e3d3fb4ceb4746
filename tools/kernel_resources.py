"""Per-kernel VGPR / SGPR / LDS / scratch figures from a `hipcc -S --cuda-device-only` listing (the .amdhsa_ blocks).
    python tools/kernel_resources.py listing.s [filter]"""
import re, subprocess, sys

def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n

def main():
    txt = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
        name = demangle(m.group(1))
        if flt and flt not in name:
            continue
        body = m.group(2)
        g = lambda k: (re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body) or [None, "?"])[1]
        print(f"{name[:110]:110s} vgpr {g('next_free_vgpr'):>4s} agpr_off {g('accum_offset'):>4s} sgpr {g('next_free_sgpr'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s}")

if __name__ == "__main__":
    main()
