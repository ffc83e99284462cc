"""HIP backend: ctypes binding (lib), kernel wrappers (kernels), block runners (blocks)."""
