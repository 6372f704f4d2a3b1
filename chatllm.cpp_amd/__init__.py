"""chatllm.cpp_amd -- MI355X-native backend for chatllm.cpp's transformer forward hot path.

Python host-side mirror of the reference's operator surface for this path: the functions in
`ops` carry the names and argument meaning of `chatllm::ggml::*` (reference src/layers.h:41-304,
src/layers.cpp:602-614,893-983,1062-1107) and call straight into the C ABI of
`libchatllm_hip.so` (include/chatllm_hip.h).  There is NO CPU fallback: if the HIP library is
missing or no GPU is visible, every op raises.

The directory name contains a dot, so import it with `load_package()` from the repo root:

    import importlib.util, sys
    spec = importlib.util.spec_from_file_location("chatllm_cpp_amd", "chatllm.cpp_amd/__init__.py",
                                                  submodule_search_locations=["chatllm.cpp_amd"])
    mod = importlib.util.module_from_spec(spec); sys.modules["chatllm_cpp_amd"] = mod; spec.loader.exec_module(mod)
"""
from . import lib as lib          # noqa: F401  (ctypes binding, loads lazily)
from .tensor import Tensor, F32, F16, Q4_0, Q4_1, Q8_0, Q4_K, Q5_K, Q6_K, I32, I64  # noqa: F401
from . import ops                 # noqa: F401
from . import synth               # noqa: F401
from .llama import Llama          # noqa: F401

__all__ = ["lib", "Tensor", "ops", "synth", "Llama", "F32", "F16", "Q4_0", "Q4_1", "Q8_0", "Q4_K", "Q5_K", "Q6_K", "I32", "I64"]
