"""Drop-in for the reference's ``multimodal_fusion/`` scripts (same module and symbol names)."""
