"""Command-line helpers (``python -m distributed_tensorflow_amd.tools.<name>``)."""
