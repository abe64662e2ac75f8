from .audio_data_loader import AudioDataLoader
