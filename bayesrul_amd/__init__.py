"""bayesrul_amd — MI355X-native SVI/ELBO hot path of lbasora/bayesrul."""
__version__ = "0.1.0"
