"""Explicit-rule model wirings (ref: lxt/explicit/models/).  The reference vendors modified copies of HF modeling files; here an
UNMODIFIED HuggingFace model instance is re-wired in place (instance-level forwards + rule modules), so the wiring follows whatever
transformers version is installed."""
