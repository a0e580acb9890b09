"""relax-vqa_amd: MI355X-native ReLaX-VQA feature-extraction hot path."""
