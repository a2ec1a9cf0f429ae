"""lush-march: MI355X-native LuSh-NeRF ray-march hot path."""
