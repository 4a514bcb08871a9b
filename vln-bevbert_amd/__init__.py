"""vln-bevbert_amd: MI355X-native BEVBert cross-modal hot path."""
__version__ = "0.1.0"
