"""cupoch.utility mirror: the device-vector wrappers users assign to a
PointCloud (`pcd.points = cph.utility.Vector3fVector(np_array)`,
`np.asarray(pcd.points.cpu())`; reference: src/python/cupoch_pybind/
device_vector_wrapper.{h,cu}, utility/eigen.cpp:123-200).  Backed by torch CUDA
tensors; DLPack is the zero-copy bridge (utility/dl_converter.cu:42-99)."""
import numpy as np

try:
    import torch
except Exception:  # pragma: no cover
    torch = None

_default_device = 0


def initialize_allocator(*args, **kwargs):
    """cupoch.initialize_allocator (src/python/cupoch/__init__.py:1-2): the
    engine owns one arena per context, nothing to configure."""
    return None


def set_default_device(index):
    global _default_device
    _default_device = int(index)


def default_device():
    return _default_device


def _to_device_tensor(a, cols, dtype):
    if torch is None or not torch.cuda.is_available():
        raise RuntimeError("cupoch_amd needs PyTorch-ROCm with a visible MI355X for device vectors")
    dev = torch.device("cuda", _default_device)
    if isinstance(a, DeviceVector):
        a = a.tensor
    if isinstance(a, torch.Tensor):
        t = a.to(device=dev, dtype=dtype)
    else:
        t = torch.as_tensor(np.ascontiguousarray(np.asarray(a)), dtype=dtype).to(dev)
    return t.reshape(-1, cols).contiguous() if cols else t.contiguous()


class DeviceVector:
    """device_vector_wrapper<T>: owns a CUDA tensor; `.cpu()` returns a numpy
    view-compatible host copy (one D2H copy, as the reference)."""
    cols = 3
    dtype = None

    def __init__(self, data=None):
        dt = self.dtype or torch.float32
        if data is None:
            data = np.zeros((0, self.cols), np.float32)
        self.tensor = _to_device_tensor(data, self.cols, dt)

    def cpu(self):
        return self.tensor.detach().cpu().numpy()

    def __len__(self):
        return int(self.tensor.shape[0])

    def size(self):
        return len(self)

    def __array__(self, dtype=None):
        a = self.cpu()
        return a.astype(dtype) if dtype is not None else a

    def __dlpack__(self, stream=None):
        return self.tensor.__dlpack__(stream=stream) if stream is not None else self.tensor.__dlpack__()

    def __dlpack_device__(self):
        return self.tensor.__dlpack_device__()

    @classmethod
    def from_dlpack(cls, capsule_or_obj):
        obj = cls.__new__(cls)
        obj.tensor = torch.from_dlpack(capsule_or_obj).reshape(-1, cls.cols).contiguous()
        return obj


class Vector3fVector(DeviceVector):
    cols = 3


class Vector2iVector(DeviceVector):
    cols = 2

    def __init__(self, data=None):
        if data is None:
            data = np.zeros((0, 2), np.int32)
        self.tensor = _to_device_tensor(data, 2, torch.int32)


class Matrix3fVector(DeviceVector):
    cols = 9

    def __init__(self, data=None):
        if data is None:
            data = np.zeros((0, 3, 3), np.float32)
        self.tensor = _to_device_tensor(data, 9, torch.float32).reshape(-1, 3, 3)

    def cpu(self):
        return self.tensor.detach().cpu().numpy()
