"""MI355X-native TensoRF VM-split ray-marching renderer (the Text2NeRF hot path)."""
