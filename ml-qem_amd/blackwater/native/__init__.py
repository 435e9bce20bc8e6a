"""ctypes binding of libmlqem_hip.so and the torch.autograd wrappers around its kernels."""
