"""Same import path as the reference backend: ``from gpu_backend.kernel_state_ansatz import ...``
works once this package directory's parent is on ``sys.path``."""
