"""Static check of the library's ISA (fab_torch_amd/_isa_check.py): no instruction touches a register whose vector-memory load
may still be in flight - the property the hand-counted `s_waitcnt vmcnt(N)` of the stream / ring kernels (stream_r8.h,
spline_r8.h, flow_r8.h, flow_r4.h, flow_device.h) rely on.  Usage: python tools/check_r8_isa.py [object stems ...]
(default: every object of the in-tree build, fab_torch_amd/build/*.o; build first).  No GPU needed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fab_torch_amd import _isa_check as chk          # noqa: E402

BUILD = os.path.join(ROOT, "fab_torch_amd", "build")
DEVICE_OBJECTS = ("flow_kernels", "ais_kernels", "spline_kernels", "train_kernels", "reduce_resample", "generic_kernels", "topk")


def main():
    stems = sys.argv[1:] or DEVICE_OBJECTS
    stems = ["spline_kernels" if s == "spline" else ("ais_kernels" if s == "flow" else s) for s in stems]
    rc = 0
    for stem in stems:
        obj = os.path.join(BUILD, stem + ".o")
        if not os.path.exists(obj):
            print(f"{obj}: missing (python -m fab_torch_amd._build)")
            rc = 1
            continue
        res = chk.check_object(obj, verbose=bool(os.environ.get("R8ISA_VERBOSE")))
        n_bad = sum(1 for v in res.values() if v)
        print(f"{stem}: {len(res)} kernels, {n_bad} with findings")
        for k, v in res.items():
            if v:
                print("  ", k[:120])
                for f in v[:8]:
                    print("     ", f)
        rc |= bool(n_bad)
    return rc


if __name__ == "__main__":
    sys.exit(main())
