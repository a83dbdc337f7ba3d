"""MI355X-native ASR hot path (front-end + encoder + decoder) behind the reference's call surface."""
