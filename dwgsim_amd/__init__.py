"""dwgsim_amd: host-side mirror (ctypes over the C-ABI) of the MI355X-native dwgsim hot path."""
