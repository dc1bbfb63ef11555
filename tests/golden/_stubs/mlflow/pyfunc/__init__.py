class PythonModel:
    pass


class PythonModelContext:
    pass


def save_model(*_, **__):
    return None
