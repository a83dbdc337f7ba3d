import ctypes as C, importlib, sys
sys.path.insert(0, "/root/repo")
lib = importlib.import_module("automatic-speech-recognition-asr-onnx_amd._lib")
l = importlib.import_module("automatic-speech-recognition-asr-onnx_amd._probe").load()
for nwg in (64, 128, 256):
    for it in (200, 2000):
        us = C.c_float(0)
        lib.check(l.asr_probe_grid_barrier(nwg, it, C.byref(us)))
        print("workgroups", nwg, "iters", it, "us/barrier", round(us.value, 2), flush=True)
for mode in (1, 2):
    for nwg in (64, 128, 256):
        us = C.c_float(0)
        lib.check(l.asr_probe_grid_barrier2(nwg, 2000, mode, C.byref(us)))
        print("hierarchical mode", mode, "workgroups", nwg, "us/barrier", round(us.value, 2), flush=True)
for kib in (128, 256, 384):
    for plain in (0, 8):
        us = C.c_float(0)
        lib.check(l.asr_probe_grid_barrier2(256, 500, 2 + plain + 16 * kib, C.byref(us)))
        print("256 workgroups, bulk", kib, "KiB per workgroup per round,", "plain" if plain else "sc1", "loads: us/round", round(2 * us.value, 2), flush=True)
