import torch
print(torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
lo = ctypes.c_int(); hi = ctypes.c_int()
print(hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)), lo.value, hi.value)
