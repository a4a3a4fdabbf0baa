"""Drop-in for the reference's ``early_fusion/`` scripts (same module and symbol names)."""
