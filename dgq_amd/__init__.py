"""dgq_amd — MI355X-native quantized-UNet inference path of DGQ."""
__version__ = "0.1.0"
