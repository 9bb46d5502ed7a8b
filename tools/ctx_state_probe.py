"""Ad-hoc: what hipDevicePrimaryCtxGetState reports before / after torch touches the device (is `active` a usable
'this device has already been used by the process' test for bp_use_blocking_sync?)."""
import ctypes as C, os, sys
import torch
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
def state(tag):
    flags, active = C.c_uint(), C.c_int()
    rc = hip.hipDevicePrimaryCtxGetState(0, C.byref(flags), C.byref(active))
    f2 = C.c_uint()
    print("%-40s rc=%d flags=%#x active=%d" % (tag, rc, flags.value, active.value), flush=True)
state("after import torch")
torch.cuda.is_available(); state("after torch.cuda.is_available()")
torch.cuda.device_count(); state("after device_count")
torch.cuda.set_device(0); state("after torch.cuda.set_device(0)")
x = torch.zeros(4, device="cuda"); state("after a tensor on cuda")
torch.cuda.synchronize(); state("after synchronize")
