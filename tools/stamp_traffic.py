"""Write profiles/<round>_traffic.json from the FETCH_SIZE / WRITE_SIZE averages of a `tools/gpu.sh traffic` run, stamped with the hash of
the kernel sources in the tree (bench.py compares it with the sources it runs on).
    python tools/stamp_traffic.py <FETCH_SIZE KiB> <WRITE_SIZE KiB> [envs] ["build note"] [round tag, default r3]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

fetch, write = float(sys.argv[1]), float(sys.argv[2])
envs = int(sys.argv[3]) if len(sys.argv) > 3 else 1048576
tag = sys.argv[5] if len(sys.argv) > 5 else 'r3'
d = {
    'build': sys.argv[4] if len(sys.argv) > 4 else 'round 3 final',
    'source_hash': bench.kernel_source_hash(),
    'envs_per_gpu': envs,
    'kernel': 'mg_kernel<3,5,true,true>',
    'FETCH_SIZE_KiB': fetch,
    'WRITE_SIZE_KiB': write,
    'hbm_bytes_per_launch': int(round((2 * fetch + write) * 1024)),
    'note': ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over tools/kbench (tools/gpu.sh traffic, profiles/%s_traffic_1M_pmc.txt), '
             'average of 54 step launches; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B read requests at 64 B); WRITE_SIZE taken '
             'as is.  source_hash = sha256 of multigrid_amd/csrc/{mgrid.hip,mg_dyn.h,mg_vis.h,mgrid_dynroll.hip} at the time of the passes.') % tag,
}
json.dump(d, open(os.path.join(ROOT, 'profiles', tag + '_traffic.json'), 'w'), indent=1)
print(d['source_hash'], d['hbm_bytes_per_launch'])
