def import_numpy():
    import numpy
    return numpy
