"""Diagnostic: how many workgroups the launch gets (rt_stats.grid_blocks; 1024 = four per CU) and cfg2's kernel time
for a given build of the library.  usage: python tools/occ_check.py path/to/lib.so"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RTIOW_HIP_LIB"] = sys.argv[1]
import rtiow_amd as rt
r = rt.Renderer(0); r.upload_scene(rt.random_scene(1).flatten())
_, _, st = r.render(rt.book1_camera(1200, 675), rt.make_params(1200, 675, 100), want_fix=False)
_, _, st = r.render(rt.book1_camera(1200, 675), rt.make_params(1200, 675, 100), want_fix=False)
print(sys.argv[1], st["grid_blocks"], st["kernel_ms"])
