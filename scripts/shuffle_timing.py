import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
from matfac_amd import synth
lib = synth._host()
lib.mfh_shuffle_check.argtypes = [C.c_int64, C.c_uint32, C.POINTER(C.c_double)]
lib.mfh_shuffle_check32.argtypes = [C.c_int64, C.c_uint32, C.POINTER(C.c_double)]
for n in (20000000, 20000000):
    secs = (C.c_double * 3)()
    print(n, lib.mfh_shuffle_check32(n, 7, secs), "std %.3f s  mfhShuffle on 32-bit entries %.3f s form %d" % (secs[0], secs[1], secs[2]), flush=True)
for n in (20000000, 20000000):
    secs = (C.c_double * 3)()
    print(n, lib.mfh_shuffle_check(n, 7, secs), "std %.3f s  mfhShuffle %.3f s form %d (MFX_SHUFFLE_THREADS=%s MFX_SHUFFLE_LIBGEN=%s)" % (secs[0], secs[1], secs[2], os.environ.get("MFX_SHUFFLE_THREADS", "2"), os.environ.get("MFX_SHUFFLE_LIBGEN", "0")), flush=True)
