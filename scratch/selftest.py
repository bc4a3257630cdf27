import sys, ctypes as C
sys.path.insert(0,'.')
from vulkanrenderer_amd import _lib
lib=_lib.load()
out=(C.c_uint64*7)()
_lib.check(lib.vkr_selftest_math(out))
print(list(out))
