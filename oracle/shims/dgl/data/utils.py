def save_info(path, info):
    raise NotImplementedError


def load_info(path):
    raise NotImplementedError
