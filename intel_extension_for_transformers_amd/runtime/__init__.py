from .engine import WoqDecoderEngine, build_rope_tables, fuse_gate_up, synth_llama_weights  # noqa: F401
