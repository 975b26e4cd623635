"""Scene record (reference Scenes/Scene.py:1-8): spheres, lights, camera,
screen and an optional ambient light."""


class Scene:
    def __init__(self, objects, light_sources, camera, screen, ambient_light=None):
        self.objects = list(objects)
        self.light_sources = list(light_sources)
        self.camera = camera
        self.screen = screen
        self.ambient_light = ambient_light
