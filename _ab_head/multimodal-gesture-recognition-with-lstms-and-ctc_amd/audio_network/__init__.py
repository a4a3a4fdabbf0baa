"""Drop-in for the reference's ``audio_network/`` scripts (same module and symbol names)."""
