class Meshes: pass
class Pointclouds: pass
