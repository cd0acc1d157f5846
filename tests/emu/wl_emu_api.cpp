// libwl_emu.so, unit 'api': the same kernel bodies and C ABI as the matching unit of libwavelets_hip.so, executed on the host.
#include "wl_backend_emu.h"
#include "../../pytorch_wavelets_amd/csrc/wl_api.inc"

// test hook (emulator only, not part of the C ABI): the size of the emulated chip, for launcher policies that depend on it
extern "C" void wl_emu_set_cus(int n) { wl_emu_cus_v = n > 0 ? n : 2; }
