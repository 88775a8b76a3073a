"""diarizen_amd — MI355X-native engine for the DiariZen sliding-window inference hot path."""
__version__ = "0.1.0"
