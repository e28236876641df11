import ctypes, sys, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmini.so"))
print("named:", lib.mini_run(1), flush=True)
print("anon:", lib.mini_run(0), flush=True)
