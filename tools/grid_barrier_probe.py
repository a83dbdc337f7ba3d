import ctypes as C, importlib, sys
sys.path.insert(0, "/root/repo")
lib = importlib.import_module("automatic-speech-recognition-asr-onnx_amd._lib")
l = lib.load()
for nwg in (64, 128, 256):
    for it in (200, 2000):
        us = C.c_float(0)
        lib.check(l.asr_debug_grid_barrier(nwg, it, C.byref(us)))
        print("workgroups", nwg, "iters", it, "us/barrier", round(us.value, 2), flush=True)
